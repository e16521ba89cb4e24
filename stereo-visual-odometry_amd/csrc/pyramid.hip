// pyramid.hip -- LK image pyramid (cv::buildOpticalFlowPyramid, called inside
// cv::calcOpticalFlowPyrLK at reference src/tracking.cpp:593-618) for gfx950.
//
// Layout in HBM: one "slot" per image holds levels 0..3, each stored WITH a kPad = 32 pixel
// BORDER_REFLECT_101 frame (upstream keeps a winSize = 21 frame the same way) and a 64-byte
// aligned row pitch, so the LK kernel gathers 4-byte aligned tiles with no bounds logic.
// Level l+1 = pyrDown(level l): separable [1 4 6 4 1], (sum + 128) >> 8, reflect-101 at the
// level's own edges -- which is exactly what reading the previous level's stored frame gives.
//
// Launch sequence per batch (all images of the batch in one grid):
//   copy0   : source image -> level-0 interior, 16 bytes per lane when alignment allows
//   border  : frame of level l from its own interior (1 load + 1 store per frame byte)
//   down    : level l interior from level l-1 (+ its frame): 4 output pixels per lane, the 5x11
//             source bytes fetched as aligned dwords, horizontal pass in registers
// Algorithmic HBM bytes per image: W*H read, 1.33*W*H written (+ frames).
// Integer-exact against oracle/lk.c (orc_pyramid_build).
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

__device__ __forceinline__ const uint8_t *src_image(const PyrArgs &a, int b)
{
    // image b of the batch: even/odd images may come from two arrays (left / right frames)
    if (a.img2) return ((b & 1) ? a.img2 : a.img) + (int64_t)(b >> 1) * a.img_stride;
    return a.img + (int64_t)b * a.img_stride;
}

__global__ __launch_bounds__(256) void pyr_copy0_kernel(PyrArgs a)
{
    // block = (lanes per row, rows): a KITTI row is 78 sixteen-byte chunks, so one row per 256-thread
    // workgroup left two thirds of it idle and the launch bound by workgroup turnaround
    const int b = blockIdx.z, y = blockIdx.y * blockDim.y + threadIdx.y;
    const int w = a.g.w[0];
    if (y >= a.g.h[0]) return;
    const uint8_t *src = src_image(a, b) + (int64_t)y * a.pitch;
    uint8_t *dst = a.slots + (int64_t)b * a.slot_stride + a.g.origin[0] + (int64_t)y * a.g.pitch[0];
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (x >= w) return;
    if (x + 16 <= w && ((uintptr_t)(src + x) & 15) == 0) {
        *(uint4 *)(dst + x) = *(const uint4 *)(src + x);          // dst rows are 32-byte aligned
    } else {
        for (int q = 0; q < 16 && x + q < w; q++) dst[x + q] = src[x + q];
    }
}

// frame of level l: every padded position outside the interior copies its reflect-101 source.
// Workgroup g < 2*kPad handles one full top/bottom frame row; the others handle four interior
// rows each (64 lanes = the 2 x 32 side bytes of a row), so no workgroup is launched for nothing.
constexpr int kBorderRows = 16;              // interior rows whose side bytes one pass of a workgroup writes
// A workgroup takes kFrameRowsPerWg full frame rows or kSidePasses x 16 interior rows: with one row / 16 rows per
// workgroup the launch was 45 k tiny workgroups per level and bound by their dispatch (58 us for 80 MB of frames)
constexpr int kFrameRowsPerWg = 4, kSidePasses = 4;
__host__ __device__ inline int border_top_groups() { return 2 * kPad / kFrameRowsPerWg; }
__host__ __device__ inline int border_side_groups(int h) { return (h + kBorderRows * kSidePasses - 1) / (kBorderRows * kSidePasses); }
__global__ __launch_bounds__(256) void pyr_border_kernel(PyrArgs a, int l)
{
    const int b = blockIdx.y;
    const int w = a.g.w[l], h = a.g.h[l], pitch = a.g.pitch[l];
    uint8_t *lvl = a.slots + (int64_t)b * a.slot_stride + a.g.origin[l];
    if ((int)blockIdx.x < border_top_groups()) {
        // full frame rows, four bytes per thread (the padded row starts 4-byte aligned)
        for (int q = 0; q < kFrameRowsPerWg; q++) {
            const int gidx = blockIdx.x * kFrameRowsPerWg + q;
            const int py = gidx < kPad ? gidx - kPad : h + (gidx - kPad);
            const uint8_t *src = lvl + (int64_t)refl101(py, h) * pitch;
            uint8_t *dst = lvl + (int64_t)py * pitch;
            for (int px = ((int)threadIdx.x << 2) - kPad; px < w + kPad; px += 1024) {
                uint32_t v = 0;
                if (px >= 0 && px + 4 <= w) v = *(const uint32_t *)(src + px);     // above / below the interior: an aligned dword of the mirrored row
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++) v |= (uint32_t)src[refl101(min(px + k, w + kPad - 1), w)] << (8 * k);
                }
                if (px + 4 <= w + kPad) *(uint32_t *)(dst + px) = v;
                else for (int k = 0; px + k < w + kPad; k++) dst[px + k] = (uint8_t)(v >> (8 * k));
            }
        }
        return;
    }
    const int sgroup = blockIdx.x - border_top_groups();
    if (w >= 2 * kPad) {
        // the 2 x 32 side bytes of sixteen interior rows per pass, a DWORD per thread (sixteen per row: 64 one-byte loads and
        // stores per row made this launch four times slower than its bytes): the four frame bytes px .. px + 3 are the
        // interior bytes at the mirrored columns in reverse order -- one unaligned load, one byte swap, one store
        static_assert(kBorderRows == 16 && kPad == 32, "256 threads = 16 rows x 16 dwords");
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        const int k = threadIdx.x & 15, r = threadIdx.x >> 4;
        // left: px = -32 + 4 k mirrors columns 32 - 4 k .. 29 - 4 k; right: px = w + 4 (k - 8) mirrors w - 2 - 4 (k - 8) .. - 3
        const int px = k < 8 ? -kPad + 4 * k : w + 4 * (k - 8);
        const int c0 = k < 8 ? kPad - 3 - 4 * k : w - 5 - 4 * (k - 8);
        for (int q = 0; q < kSidePasses; q++) {
            const int py = (sgroup * kSidePasses + q) * kBorderRows + r;
            if (py >= h) return;
            uint8_t *row = lvl + (int64_t)py * pitch;
            const uint32_t v = *(const u32_unaligned *)(row + c0);
            *(u32_unaligned *)(row + px) = __builtin_amdgcn_perm(0u, v, 0x00010203u);
        }
    } else {
        const int t = threadIdx.x & 63;
        const int px = t < kPad ? t - kPad : w + (t - kPad);
        const int sx = refl101(px, w);
        for (int q = 0; q < kSidePasses; q++)
            for (int r = threadIdx.x >> 6; r < kBorderRows; r += 4) {
                const int py = (sgroup * kSidePasses + q) * kBorderRows + r;
                if (py >= h) return;
                uint8_t *row = lvl + (int64_t)py * pitch;
                row[px] = row[sx];
            }
    }
}

// level l (>= 1) interior from level l-1: thread -> 4 consecutive output pixels
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrArgs a, int l)
{
    const int b = blockIdx.z, y = blockIdx.y * blockDim.y + threadIdx.y;     // block = (lanes per row, rows)
    const int w = a.g.w[l];
    if (y >= a.g.h[l]) return;
    const int sp = a.g.pitch[l - 1];
    uint8_t *slot = a.slots + (int64_t)b * a.slot_stride;
    const uint8_t *src = slot + a.g.origin[l - 1];
    uint8_t *dst = slot + a.g.origin[l] + (int64_t)y * a.g.pitch[l];
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x0 >= w) return;
    // source columns 2*x0-2 .. 2*x0+8, fetched as the 4 aligned dwords starting at 2*x0-4
    // (x0 % 4 == 0 and the level origin is 32-byte aligned, so 2*x0-4 is 4-byte aligned; the
    //  stored reflect-101 frame supplies columns < 0 and >= w_src, rows likewise)
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint32_t *row = (const uint32_t *)(src + (int64_t)(2 * y + r - 2) * sp + 2 * x0 - 4);
        const uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3];
        // bytes s[j] = source column 2*x0 - 4 + j, j = 0..15
        int s[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            s[j] = (d0 >> (8 * j)) & 255; s[4 + j] = (d1 >> (8 * j)) & 255;
            s[8 + j] = (d2 >> (8 * j)) & 255; s[12 + j] = (d3 >> (8 * j)) & 255;
        }
        const int wv = (r == 2) ? 6 : ((r == 1 || r == 3) ? 4 : 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = 4 + 2 * q;                       // centre column 2*(x0+q)
            const int hs = s[c] * 6 + (s[c - 1] + s[c + 1]) * 4 + s[c - 2] + s[c + 2];
            acc[q] += hs * wv;
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) out |= (uint32_t)((acc[q] + 128) >> 8) << (8 * q);
    if (x0 + 4 <= w) *(uint32_t *)(dst + x0) = out;
    else for (int q = 0; x0 + q < w; q++) dst[x0 + q] = (uint8_t)(out >> (8 * q));
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
    // a block of up to 256 threads = (lanes a row needs, as many rows as fit)
    auto shape = [](int lanes_per_row, int rows, dim3 &grid, dim3 &block, int batch) {
        const int bx = lanes_per_row < 256 ? lanes_per_row : 256, by = 256 / bx;
        block = dim3(bx, by, 1);
        grid = dim3((lanes_per_row + bx - 1) / bx, (rows + by - 1) / by, batch);
    };
    {
        dim3 g, bk;
        shape((a.g.w[0] + 15) / 16, a.g.h[0], g, bk, batch);
        hipLaunchKernelGGL(pyr_copy0_kernel, g, bk, 0, st, a);
    }
    for (int l = 0; l < a.g.nlevels; l++) {
        if (l > 0) {
            dim3 g, bk;
            shape((a.g.w[l] + 3) / 4, a.g.h[l], g, bk, batch);
            hipLaunchKernelGGL(pyr_down_kernel, g, bk, 0, st, a, l);
        }
        dim3 gb(border_top_groups() + border_side_groups(a.g.h[l]), batch, 1);
        hipLaunchKernelGGL(pyr_border_kernel, gb, blk, 0, st, a, l);
    }
}

void launch_pyr_read_level(const PyrGeom &g, const uint8_t *slot, int l, uint8_t *out, int out_pitch,
                           hipStream_t st)
{
    dim3 grid((g.w[l] + 255) / 256, g.h[l], 1), blk(256, 1, 1);
    hipLaunchKernelGGL(pyr_read_level_kernel, grid, blk, 0, st, g, slot, l, out, out_pitch);
}

}  // namespace svo
