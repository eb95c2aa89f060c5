// pcx_crop.hip -- device croppers (cropping.py).  Placeholder entry points;
// the kernels land with SURVEY.md section 8 rows a15-a17.
#include "pcx_internal.h"

using pcx::set_error;

struct pcx_cropper {};

extern "C" {
int pcx_cropper_create(pcx_engine*, const pcx_cropper_desc*, pcx_cropper**) {
  return set_error(PCX_E_UNSUPPORTED, "pcx_cropper_create: croppers are not built yet");
}
void pcx_cropper_destroy(pcx_cropper*) {}
int pcx_cropper_crop(pcx_cropper*, void*) { return set_error(PCX_E_UNSUPPORTED, "croppers are not built yet"); }
int pcx_cropper_buffers(pcx_cropper*, uint8_t**, int32_t**) {
  return set_error(PCX_E_UNSUPPORTED, "croppers are not built yet");
}
}
