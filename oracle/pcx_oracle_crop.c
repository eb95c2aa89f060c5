/*
 * pcx_oracle_crop.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see pcx_oracle.h).
 * Scalar restatement of pycolab/cropping.py: _do_crop (:118-227),
 * FixedCropper.crop (:255-268), ScrollingCropper (:271-598).
 *
 * A cropper holds one window state per environment; an environment whose
 * episode restarted gets a fresh window, as a cropper does when set_engine()
 * hands it a new Engine (:378-391).
 */
#include "pcx_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* accessors implemented in pcx_oracle.c */
int pcxo__rows(const pcxo_engine* e);
int pcxo__cols(const pcxo_engine* e);
int pcxo__n_chars(const pcxo_engine* e);
int pcxo__char(const pcxo_engine* e, int k);
int64_t pcxo__batch(const pcxo_engine* e);
const uint8_t* pcxo__planes(const pcxo_engine* e, int64_t b);
int pcxo__frame(const pcxo_engine* e, int64_t b);
/* centroid of entity `ch` in env b (cropping.py:544-598); 0 if it has none */
int pcxo__centroid(const pcxo_engine* e, int64_t b, int ch, int* row, int* col);
int pcxo__valid_char(const pcxo_engine* e, int ch);

struct pcxo_cropper {
  pcxo_engine* e;
  pcx_cropper_desc d;
  int out_cells;
  uint8_t* planes;   /* [batch][1+L][rows*cols] */
  int32_t* corner;   /* [batch][2] */
  uint8_t* has_corner;
  uint8_t* error;    /* [batch]: 1 = the reference would raise */
};

int pcxo_cropper_create(pcxo_engine* e, const pcx_cropper_desc* d, pcxo_cropper** out) {
  if (!e || !d || !out || d->rows <= 0 || d->cols <= 0) return PCX_E_INVALID;
  if (d->pad_char >= 0 && !pcxo__valid_char(e, d->pad_char)) return PCX_E_INVALID; /* :186-188 ValueError */
  if (d->kind == PCX_CROP_SCROLLING && d->pad_char < 0 &&
      (pcxo__rows(e) < d->rows || pcxo__cols(e) < d->cols)) return PCX_E_INVALID;  /* :381-390 */
  pcxo_cropper* c = (pcxo_cropper*)calloc(1, sizeof *c);
  c->e = e;
  c->d = *d;
  c->out_cells = d->rows * d->cols;
  int64_t B = pcxo__batch(e);
  c->planes = (uint8_t*)calloc((size_t)B * (1 + pcxo__n_chars(e)), c->out_cells);
  c->corner = (int32_t*)calloc((size_t)B * 2, sizeof(int32_t));
  c->has_corner = (uint8_t*)calloc((size_t)B, 1);
  c->error = (uint8_t*)calloc((size_t)B, 1);
  *out = c;
  return 0;
}

void pcxo_cropper_destroy(pcxo_cropper* c) {
  if (!c) return;
  free(c->planes); free(c->corner); free(c->has_corner); free(c->error);
  free(c);
}

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

/* cropping.py:118-227 */
static void do_crop(pcxo_cropper* c, int64_t b, int top, int left) {
  const pcxo_engine* e = c->e;
  int R = pcxo__rows(e), C = pcxo__cols(e), L = pcxo__n_chars(e);
  int cr = c->d.rows, cc = c->d.cols, n = cr * cc;
  int bottom = top + cr, right = left + cc;
  uint8_t* out = c->planes + (size_t)b * (1 + L) * n;
  const uint8_t* in = pcxo__planes(e, b);
  if (c->d.pad_char < 0) {
    if (top < 0 || left < 0 || bottom > R || right > C) { c->error[b] = 1; return; } /* :175-183 */
  } else {
    memset(out, c->d.pad_char, n);                                                 /* :189 */
    for (int k = 0; k < L; ++k) memset(out + (size_t)(1 + k) * n, c->d.pad_char == pcxo__char(e, k), n); /* :190-191 */
  }
  int from_tr = imax(0, top), from_lc = imax(0, left);
  int from_bre = imax(0, imin(R, bottom)), from_rce = imax(0, imin(C, right));
  int to_tr = imax(0, -top), to_lc = imax(0, -left);
  int rows = from_bre - from_tr, cols = from_rce - from_lc;
  if (rows <= 0 || cols <= 0) return; /* fully outside: pure padding */
  for (int p = 0; p < 1 + L; ++p)
    for (int r = 0; r < rows; ++r)
      memcpy(out + (size_t)p * n + (to_tr + r) * cc + to_lc,
             in + (size_t)p * R * C + (from_tr + r) * C + from_lc, cols);
}

/* cropping.py:539-542 */
static void rectify(pcxo_cropper* c, int32_t* corner) {
  int R = pcxo__rows(c->e), C = pcxo__cols(c->e);
  corner[0] = imax(0, corner[0]) - imax(0, corner[0] + c->d.rows - R);
  corner[1] = imax(0, corner[1]) - imax(0, corner[1] + c->d.cols - C);
}

/* cropping.py:438-458 */
static void initialise(pcxo_cropper* c, int32_t* corner, int have, int crow, int ccol, int off_r, int off_c) {
  if (!have) { corner[0] = corner[1] = 0; return; }
  corner[0] = crow - off_r;
  corner[1] = ccol - off_c;
  if (c->d.pad_char < 0) rectify(c, corner);
}

int pcxo_cropper_crop(pcxo_cropper* c) {
  const pcxo_engine* e = c->e;
  int64_t B = pcxo__batch(e);
  int R = pcxo__rows(e), C = pcxo__cols(e);
  for (int64_t b = 0; b < B; ++b) {
    c->error[b] = 0;
    if (c->d.kind == PCX_CROP_FIXED) { do_crop(c, b, c->d.top, c->d.left); continue; }
    if (pcxo__frame(e, b) == 0) c->has_corner[b] = 0; /* a new episode == a new Engine */
    int32_t* corner = c->corner + 2 * b;
    int crow = 0, ccol = 0, have = 0;
    for (int i = 0; i < c->d.n_track && !have; ++i) /* :544-549 */
      have = pcxo__centroid(e, b, c->d.to_track[i], &crow, &ccol);
    int rows = c->d.rows, cols = c->d.cols, mrow = c->d.margin_rows, mcol = c->d.margin_cols;
    if (!c->has_corner[b]) { /* :399-402 */
      initialise(c, corner, have, crow, ccol, rows / 2 + c->d.initial_offset_rows, cols / 2 + c->d.initial_offset_cols);
      c->has_corner[b] = 1;
    } else if (have) {
      int wrow = corner[0], wcol = corner[1];
      int can_vert = (mrow - 1) <= (crow - wrow) && (crow - wrow) <= (rows - mrow);   /* :484 */
      int can_horiz = (mcol - 1) <= (ccol - wcol) && (ccol - wcol) <= (cols - mcol);  /* :487 */
      if (c->d.pad_char < 0) { /* :491-504 */
        if (!can_vert) {
          if (wrow <= 0) can_vert = crow <= mrow;
          else if (wrow >= R - rows) can_vert = crow >= wrow + rows - mrow;
        } else if (!can_horiz) {
          if (wcol <= 0) can_horiz = ccol <= mcol;
          else if (wcol >= C - cols) can_horiz = ccol >= wcol + cols - mcol;
        }
      }
      if (can_vert && can_horiz) { /* _pan_to :508-534 */
        int drow = imin(0, crow - wrow - mrow), dcol = imin(0, ccol - wcol - mcol);
        if (drow == 0) drow += imax(0, crow - wrow - rows + mrow + 1);
        if (dcol == 0) dcol += imax(0, ccol - wcol - cols + mcol + 1);
        corner[0] = wrow + drow;
        corner[1] = wcol + dcol;
        if (c->d.pad_char < 0) rectify(c, corner);
      } else if (c->d.saccade) { /* :414-415 */
        initialise(c, corner, 1, crow, ccol, rows / 2, cols / 2);
      }
    }
    do_crop(c, b, corner[0], corner[1]);
  }
  return 0;
}

int pcxo_cropper_buffers(pcxo_cropper* c, uint8_t** planes, int32_t** corner) {
  if (planes) *planes = c->planes;
  if (corner) *corner = c->corner;
  return 0;
}
const uint8_t* pcxo_cropper_errors(pcxo_cropper* c) { return c->error; }
