// Register-stationary 5x5 / stride-2 convolution on 64 input channels (conv_wstat.hip), reached through ft_conv_direct_*.
#pragma once
#include "ft_common.h"

namespace ft {

struct WsPlan {
  int ncg;               // output-channel groups of 64 (each workgroup keeps ONE group's weights in registers for its lifetime)
  int tiles_x, tiles_y;  // 8 x 8 output patches per image row / column
  int npatches;          // N * tiles_y * tiles_x
};

int ws_plan(const ft_conv_desc* d, WsPlan* out);   // FT_OK where the form applies
long long ws_weight_bytes(const WsPlan& pl);
int ws_pack(const ft_conv_desc* d, const WsPlan& pl, const void* w_packed, int kpad, int cout_pad, void* wstream, hipStream_t stream);
int ws_launch(const ft_conv_desc* d, const WsPlan& pl, const void* x, const void* wstream, const float* scale, const float* shift,
              void* y, hipStream_t stream);

}  // namespace ft
