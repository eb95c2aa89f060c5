"""`the_plot.log` is host-only string traffic (reference: protocols/logging.py).

Device programs do not produce strings; `consume` always returns an empty list.
"""


def log(the_plot, message):
  del the_plot, message


def consume(the_plot):
  del the_plot
  return []
