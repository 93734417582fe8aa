// ofl_scatter.hip -- K3 scattered -> regular grid interpolation (placeholder until the kernel lands).
#include "ofl_common.h"
using namespace ofl;
extern "C" {
int ofl_scatter_workspace_bytes(int H, int W, int C, size_t *bytes)
{
    (void)H; (void)W; (void)C;
    if (bytes) *bytes = 0;
    return fail(OFL_E_INVALID, "ofl_scatter_linear: not implemented in this build");
}
int ofl_scatter_linear_dev(const float *, int, const uint8_t *, const float *, int, int, int,
                           const float *, float *, uint8_t *, void *, size_t, void *)
{
    return fail(OFL_E_INVALID, "ofl_scatter_linear: not implemented in this build");
}
int ofl_scatter_linear(const float *, int, const uint8_t *, const float *, int, int, int,
                       const float *, float *, uint8_t *)
{
    return fail(OFL_E_INVALID, "ofl_scatter_linear: not implemented in this build");
}
}
