/*
 * pcx_oracle_crop.c -- TEST INFRASTRUCTURE, NOT PRODUCT (see pcx_oracle.h).
 * Scalar restatement of pycolab/cropping.py (croppers).  Filled in with the
 * cropper row of SURVEY.md section 8 (a15-a17).
 */
#include "pcx_oracle.h"

#include <stdio.h>

int pcxo_cropper_create(pcxo_engine* e, const pcx_cropper_desc* d, pcxo_cropper** out) {
  (void)e; (void)d; (void)out;
  return PCX_E_UNSUPPORTED;
}
void pcxo_cropper_destroy(pcxo_cropper* c) { (void)c; }
int pcxo_cropper_crop(pcxo_cropper* c) { (void)c; return PCX_E_UNSUPPORTED; }
int pcxo_cropper_buffers(pcxo_cropper* c, uint8_t** planes, int32_t** corner) {
  (void)c; (void)planes; (void)corner;
  return PCX_E_UNSUPPORTED;
}
