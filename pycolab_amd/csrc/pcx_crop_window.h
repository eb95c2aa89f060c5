// pcx_crop_window.h -- how a cropper's window moves, one environment, shared by
// the stand-alone cropper kernel (pcx_crop.hip: pcx_crop_update) and by the step
// kernels that run fused croppers in their own logic phase (pcx_stream.h).
// Reference: pycolab/cropping.py ScrollingCropper.crop :393-426, _initialise
// :438-458, _can_pan_to :460-506, _pan_to :508-534, _rectify :536-542.
// gfx950 only.
#pragma once

#include "pcx_device.h"

namespace pcx {
namespace crop {

struct WindowRule {
  int32_t rows, cols;                // window size
  int32_t R, C;                      // observation size
  int32_t margin_rows, margin_cols;  // scroll margins
  int32_t off_rows, off_cols;        // initial_offset
  int32_t saccade;
  int32_t pad_char;                  // < 0: no pad character (the window stays on the observation)
};

// One crop() of a ScrollingCropper whose tracked entity is at (crow, ccol) when
// `have` (the first visible sprite / nonempty drape of to_track, :544-549).
// (wrow, wcol) is the window's corner, meaningful while has_corner.
__device__ __forceinline__ void move_window(const WindowRule& p, bool have, int crow, int ccol, bool& has_corner,
                                            int& wrow, int& wcol) {
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int rows = p.rows, cols = p.cols, mrow = p.margin_rows, mcol = p.margin_cols;
  auto rectify = [&]() {  // :539-542
    wrow = imax(0, wrow) - imax(0, wrow + rows - p.R);
    wcol = imax(0, wcol) - imax(0, wcol + cols - p.C);
  };
  auto initialise = [&](int off_r, int off_c) {  // :438-458
    if (!have) { wrow = 0; wcol = 0; return; }
    wrow = crow - off_r;
    wcol = ccol - off_c;
    if (p.pad_char < 0) rectify();
  };
  if (!has_corner) {
    initialise(rows / 2 + p.off_rows, cols / 2 + p.off_cols);
    has_corner = true;
  } else if (have) {
    bool can_vert = (mrow - 1) <= (crow - wrow) && (crow - wrow) <= (rows - mrow);
    bool can_horiz = (mcol - 1) <= (ccol - wcol) && (ccol - wcol) <= (cols - mcol);
    if (p.pad_char < 0) {  // :491-504, including the `elif not can_horiz`
      if (!can_vert) {
        if (wrow <= 0) can_vert = crow <= mrow;
        else if (wrow >= p.R - rows) can_vert = crow >= wrow + rows - mrow;
      } else if (!can_horiz) {
        if (wcol <= 0) can_horiz = ccol <= mcol;
        else if (wcol >= p.C - cols) can_horiz = ccol >= wcol + cols - mcol;
      }
    }
    if (can_vert && can_horiz) {  // _pan_to
      int drow = imin(0, crow - wrow - mrow), dcol = imin(0, ccol - wcol - mcol);
      if (drow == 0) drow += imax(0, crow - wrow - rows + mrow + 1);
      if (dcol == 0) dcol += imax(0, ccol - wcol - cols + mcol + 1);
      wrow += drow;
      wcol += dcol;
      if (p.pad_char < 0) rectify();
    } else if (p.saccade) {
      initialise(rows / 2, cols / 2);
    }
  }
}

// cropping.py:175-183: without a pad character the window must lie on the observation
__device__ __forceinline__ bool window_leaves_observation(const WindowRule& p, int top, int left) {
  return p.pad_char < 0 && (top < 0 || left < 0 || top + p.rows > p.R || left + p.cols > p.C);
}

// ---- croppers fused into a step kernel (include/pcx.h pcx_engine_fuse_croppers) ----
// The step kernel has the frame it paints in LDS: it moves the windows in its
// logic phase (lane == environment) and streams the cropped planes itself --
// _do_crop (cropping.py:118-227) without a second pass over the observation.
constexpr int MAX_FUSED_CROPPERS = 4;
constexpr int MAX_FUSED_TRACK = 4;
constexpr int MAX_FUSED_FEATURES = 16;  // layers of a window's fused feature stack

struct FusedWindow {
  uint8_t* out;         // [batch][1 + L][out_pitch]: the cropper's output planes
  int32_t* corner;      // [batch][2]
  uint8_t* has_corner;  // [batch]
  uint8_t* error;       // [batch]
  WindowRule rule;
  int32_t scrolling;    // 0: FixedCropper at (top, left)
  int32_t top, left;
  int32_t n_track;      // ScrollingCropper.to_track, in priority order (cropping.py:544-558)
  int32_t track_sprite[MAX_FUSED_TRACK];  // template sprite index, or the drape's index where track_kind is 1
  int32_t track_kind[MAX_FUSED_TRACK];    // 0 sprite (its position while visible), 1 drape (median of its curtain, :590-598)
  int32_t out_pitch;    // bytes per output plane (rows * cols rounded up to 4)
  uint32_t pad_planes;  // bit k: layer k (plane 1 + k) is 1 where the pad character fills (pad_char == chars[k])
  // crop -> post-process in the same launch (pcx_cropper_set_features; human_ui.py:252-265, better_scrolly_maze.py:
  // 237-247 -> rendering.py:545-661): the float32 stack of `feat_depth` layers OF THE WINDOW, [batch][depth][rows x cols]
  // (feat_hwc: [batch][rows x cols][depth]), written by the loop that has the window's board dword in a register.
  // feat_skip: 1 the window's uint8 layer planes are not written, 2 nor its board plane.  Kernels built on
  // pcx_stream.h stream_windows only (Backend::fused_window_features).
  float* feat;
  int32_t feat_depth, feat_hwc, feat_skip;
  uint8_t feat_ch[MAX_FUSED_FEATURES];
};

struct FusedCrops {
  int32_t n;     // windows in use
  int32_t only;  // the full-board planes are not written any more
  int32_t drapes;  // some window follows a drape (tracks_drapes): kernels that export the raw curtains from all waves of a
                   // cooperative workgroup move their windows after that export instead of in the logic phase
  FusedWindow w[MAX_FUSED_CROPPERS];
};

// (host) does any window follow a drape?  Only kernels that keep the raw curtains as bit rows can.
inline bool tracks_drapes(const FusedCrops* fc) {
  if (!fc) return false;
  for (int i = 0; i < fc->n; ++i)
    for (int j = 0; j < fc->w[i].n_track; ++j)
      if (fc->w[i].track_kind[j]) return true;
  return false;
}

}  // namespace crop
}  // namespace pcx
