// Activation and loss derivative of one sample, shared by the batch-side kernels (kernels_wide.hip, wide_pipe_device.h).
#pragma once
#include "tnml_internal.h"

namespace tnml {

// ------------------------------------------------------------------------------------------
// activation + loss derivative of one sample (Network_class.py:767-835).  fa and g are written in
// place over L values held in LDS at stride `st`.
// ------------------------------------------------------------------------------------------
__device__ inline void act_and_lossder(const float *fin, int st_in, float *fa, float *g, int st, int L,
                                       int y, int act_fn, int loss_fn, float T, float &sumabs,
                                       int &correct, int &nonfinite) {
  // bit 8 of act_fn: the input already went through the activation (compute_loss_derivate's
  // argument); the low bits still select the cross-entropy formula (Network_class.py:826-830)
  const bool pre_activated = (act_fn & 0x100) != 0;
  act_fn &= 0xff;
  // activation
  if (pre_activated) {
    for (int l = 0; l < L; ++l) fa[l * st] = fin[l * st_in];
  } else if (act_fn == TNML_ACT_SOFTMAX) {
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l) mx = fmaxf(mx, fin[l * st_in]);
    float sum = 0.f;
    for (int l = 0; l < L; ++l) {
      const float e = __expf((fin[l * st_in] - mx) / T);
      fa[l * st] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int l = 0; l < L; ++l) fa[l * st] *= inv;
  } else if (act_fn == TNML_ACT_SIGMOID) {
    for (int l = 0; l < L; ++l) fa[l * st] = 1.0f / (1.0f + __expf(-fin[l * st_in] / T));
  } else {
    for (int l = 0; l < L; ++l) fa[l * st] = fin[l * st_in];
  }
  // metrics: argmax (first maximum, as np.argmax) and sum |y - fa|.  All three activations are
  // monotonic, so the argmax is taken on f itself: identical in exact arithmetic, and immune to the
  // float32 saturation of sigmoid/softmax that would create ties the float64 reference does not see.
  int am = 0;
  float best = fin[0];
  float sa = 0.f;
  for (int l = 0; l < L; ++l) {
    const float v = fa[l * st];
    const float fv = fin[l * st_in];
    if (fv > best) { best = fv; am = l; }
    sa += fabsf((l == y ? 1.0f : 0.0f) - v);
    if (!isfinite(v)) nonfinite = 1;
  }
  sumabs = sa;
  correct = (am == y) ? 1 : 0;
  // loss derivative
  for (int l = 0; l < L; ++l) {
    const float v = fa[l * st];
    const float yy = (l == y) ? 1.0f : 0.0f;
    float d;
    if (loss_fn == TNML_LOSS_MSE) {
      d = yy - v;
    } else if (loss_fn == TNML_LOSS_CROSS_ENTROPY) {
      d = (act_fn == TNML_ACT_SOFTMAX) ? (yy - yy * v) / T : yy / v;
    } else {
      d = 1.0f / ((l == y ? v : v - 1.0f) + 1e-4f);
    }
    g[l * st] = d;
  }
}

}  // namespace tnml
