// pyramid.hip -- LK image pyramid (cv::buildOpticalFlowPyramid, called inside
// cv::calcOpticalFlowPyrLK at reference src/tracking.cpp:593-618) for gfx950.
//
// Layout in HBM: one "slot" per image holds levels 0..3, each stored WITH a kPad = 32 pixel
// BORDER_REFLECT_101 frame (upstream keeps a winSize = 21 frame the same way) and a 64-byte
// aligned row pitch, so the LK kernel gathers 4-byte aligned tiles with no bounds logic.
// Level l+1 = pyrDown(level l): separable [1 4 6 4 1], (sum + 128) >> 8, reflect-101 at the
// level's own edges.  Border pixels are produced by evaluating the same expression at the
// reflected interior coordinate, so every output byte is written exactly once and no second
// border-fill pass (and no inter-workgroup dependency) is needed.
// Integer-exact against oracle/lk.c (orc_pyramid_build).
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

// level 0: padded copy of the source image
__global__ __launch_bounds__(256) void pyr_level0_kernel(PyrArgs a)
{
    const int b = blockIdx.z;
    const int w = a.g.w[0], h = a.g.h[0];
    const int px = blockIdx.x * 256 + threadIdx.x;       // padded x
    const int py = blockIdx.y;                           // padded y
    if (px >= w + 2 * kPad) return;
    const uint8_t *img = a.img + (int64_t)b * a.img_stride;
    uint8_t *dst = a.slots + (int64_t)b * a.slot_stride + a.g.origin[0];
    int x = refl101(px - kPad, w), y = refl101(py - kPad, h);
    dst[(int64_t)(py - kPad) * a.g.pitch[0] + (px - kPad)] = img[(int64_t)y * a.pitch + x];
}

// level l (>= 1) from level l-1, including the border ring
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrArgs a, int l)
{
    const int b = blockIdx.z;
    const int w = a.g.w[l], h = a.g.h[l];
    const int sw = a.g.w[l - 1], sh = a.g.h[l - 1], sp = a.g.pitch[l - 1];
    const int px = blockIdx.x * 256 + threadIdx.x;
    const int py = blockIdx.y;
    if (px >= w + 2 * kPad) return;
    uint8_t *slot = a.slots + (int64_t)b * a.slot_stride;
    const uint8_t *src = slot + a.g.origin[l - 1];
    uint8_t *dst = slot + a.g.origin[l];
    const int x = refl101(px - kPad, w), y = refl101(py - kPad, h);
    // columns 2x-2..2x+2 / rows 2y-2..2y+2 of the source level, reflect-101 on ITS size
    int cx[5], acc = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) cx[k] = refl101(2 * x + k - 2, sw);
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint8_t *row = src + (int64_t)refl101(2 * y + r - 2, sh) * sp;
        int hs = row[cx[2]] * 6 + (row[cx[1]] + row[cx[3]]) * 4 + row[cx[0]] + row[cx[4]];
        const int wv = (r == 2) ? 6 : ((r == 1 || r == 3) ? 4 : 1);
        acc += hs * wv;
    }
    dst[(int64_t)(py - kPad) * a.g.pitch[l] + (px - kPad)] = (uint8_t)((acc + 128) >> 8);
}

__global__ __launch_bounds__(256) void pyr_read_level_kernel(PyrGeom g, const uint8_t *slot, int l,
                                                              uint8_t *out, int out_pitch)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.w[l]) return;
    out[(int64_t)y * out_pitch + x] = slot[g.origin[l] + (int64_t)y * g.pitch[l] + x];
}

void launch_pyramid(const PyrArgs &a, int batch, hipStream_t st)
{
    dim3 blk(256, 1, 1);
    {
        dim3 g((a.g.w[0] + 2 * kPad + 255) / 256, a.g.h[0] + 2 * kPad, batch);
        hipLaunchKernelGGL(pyr_level0_kernel, g, blk, 0, st, a);
    }
    for (int l = 1; l < a.g.nlevels; l++) {
        dim3 g((a.g.w[l] + 2 * kPad + 255) / 256, a.g.h[l] + 2 * kPad, batch);
        hipLaunchKernelGGL(pyr_down_kernel, g, blk, 0, st, a, l);
    }
}

void launch_pyr_read_level(const PyrGeom &g, const uint8_t *slot, int l, uint8_t *out, int out_pitch,
                           hipStream_t st)
{
    dim3 grid((g.w[l] + 255) / 256, g.h[l], 1), blk(256, 1, 1);
    hipLaunchKernelGGL(pyr_read_level_kernel, grid, blk, 0, st, g, slot, l, out, out_pitch);
}

}  // namespace svo
