// Register-stationary persistent-strip form of the 64-plane identity Bottleneck (bottleneck_rstat.hip), reached through
// ft_bottleneck_fwd (bottleneck.hip) where bnr_plan accepts the shape; FT_BNK_RSTAT=0 keeps the patch kernel.
#pragma once
#include "ft_common.h"

namespace ft {

struct BnrPlan {
  int SR;   // rows per strip
  int S;    // strips per image
};

// FT_OK when the strip form covers `d` (fp16 identity block, C = 256, P = 64, 3 <= W <= 62, at least eight 64-pixel steps per strip)
int bnr_plan(const ft_bottleneck_desc* d, BnrPlan* out);
int bnr_launch(const ft_bottleneck_desc* d, const BnrPlan& pl, const void* x, const void* w1, const void* w2, const void* w3,
               const float* scale_shift, void* y, hipStream_t stream);

}  // namespace ft
