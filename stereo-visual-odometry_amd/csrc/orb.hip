// orb.hip -- the reference's ORB path on gfx950 (BASELINE config #3):
//   ORBextractor::operator() (reference src/ORBextractor.cpp:990-1055): 8-level bilinear pyramid
//   (:1061-1085), per-cell FAST with threshold fallback (:717-807), quadtree distribution
//   (:430-485, 487-715), intensity-centroid orientation (:21-48), 7x7 sigma-2 blur + rotated BRIEF
//   (:51-97, 981-988), and Tracking::ORB_Robust_Find_MuliImage_MatchedFeatures
//   (src/tracking.cpp:534-581) with its BruteForce-Hamming matcher.
// Everything is integer or single-precision with the reference's operation order (FP contraction
// off), so keypoints, angles, descriptors and matches are bit-identical to oracle/orb.c.
//
// Launch sequence for a batch of images (every kernel covers all images of the batch):
//   [copy0 -- only where level 0 cannot be read in place from the input frames] -> resize(l) x7 (row-streaming) -> blur (all
//   levels) -> cellfast (all levels) -> gather -> distribute -> orient+describe -> assemble.  (No frame is written around the ORB levels: the blur, its only
//   reader, reflects at the edges itself.)
// The quadtree (std::list / sort / pointer code in the reference) is restated as an array-based
// doubly linked list in LDS walked by ONE WAVE per (image, level): the walk is serial by nature (each
// split decision depends on the node count so far), so its control flow is wave-uniform, while every
// operation on keys (DivideNode's 4-way partition, the size sort, the best-response pick) is 64 lanes
// wide; 16 instances per frame run concurrently, four per CU.
// Capacity: a cell with more than kCellCap candidates, a level with more than 4 * max_keypoints, a
// quadtree that outgrows its node pool or an image with more than max_keypoints keypoints raises
// that IMAGE's bit in orb_overflow[]; the stage call reports it as an error, the fused paths fail
// the pairs that use the image with SVO_FAIL_CAPACITY (never a silently truncated set).
#include <cstring>
#include <vector>
#include "svo_ctx.h"
#include "orb_pattern.h"

namespace svo {

__constant__ signed char c_pattern[1024];

// ---- pyramid ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void orb_copy0_kernel(OrbGeom g, const uint8_t *img, int pitch, int64_t img_stride,
                                                        uint8_t *slots, int64_t slot_stride)
{
    // block = (16-byte lanes a row needs, rows that fit 256 threads): one row per workgroup left most of it idle
    const int b = blockIdx.z, y = blockIdx.y * blockDim.y + threadIdx.y, x = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const int w = g.w[0];
    if (x >= w || y >= g.h[0]) return;
    const uint8_t *src = img + (int64_t)b * img_stride + (int64_t)y * pitch + x;
    uint8_t *dst = slots + (int64_t)b * slot_stride + g.origin[0] + (int64_t)y * g.pitch[0] + x;   // 16-byte aligned
    if (x + 16 <= w && ((uintptr_t)src & 15) == 0) *(uint4 *)dst = *(const uint4 *)src;
    else for (int q = 0; q < 16 && x + q < w; q++) dst[q] = src[q];
}

// One launch serves ALL pyramid levels where the per-level work is independent (blur, cell FAST):
// eight short launches per step each paid their own ramp-up and tail.  The level of a block is the last
// one whose first block is not beyond it (wave-uniform: a scalar loop over <= 8 entries).
constexpr int kBlurRowsPerThread = 28;

// XCD-aware (image, block of the image) from a one-dimensional block id.  Consecutive workgroup ids go round-robin
// to the 8 XCDs, each with its own 4 MB L2; dealt out in plain order an image's workgroups land on all eight, and
// every XCD fetches the image's 128-byte lines again (cell FAST: 3.6 GB of HBM traffic per 514 images against
// 0.74 GB of pixels, 14 % L2 hit rate; descriptors: 5.3 GB, the kernel ran at the HBM roof).  Images in whole
// groups of eight go one per XCD -- an image's pyramid + blurred copy (2.9 MB) then stays in ITS L2 --, the last
// n_img % 8 images are spread in the plain order.
__device__ __forceinline__ void xcd_image_block(int lin, int per_img, int n_img, int &img, int &blk)
{
    const int n_aware = (n_img & ~7) * per_img;
    if (lin < n_aware) {
        const int xcd = lin & 7, q = lin >> 3;
        const int grp = q / per_img;
        img = grp * 8 + xcd; blk = q - grp * per_img;
    } else {
        const int r = lin - n_aware, i = r / per_img;
        img = (n_img & ~7) + i; blk = r - i * per_img;
    }
}

__device__ __forceinline__ int level_of_block(const int *first, int nlevels, int blk)
{
    int l = 0;
    for (int k = 1; k < nlevels; k++) l = blk >= first[k] ? k : l;
    return l;
}

// cv::resize(level l-1 -> level l, INTER_LINEAR), 8-bit fixed point (11-bit coefficients)
// cv::resize(INTER_LINEAR) of level l-1 into level l (ComputePyramid, ORBextractor.cpp:1061-1085).
// The per-column (sx, alpha) and per-row (sy0, sy1, beta) tables are built once on the host
// (orb_make_tables: upstream builds the same tables per call), so a thread is four byte loads and
// the 11-bit fixed-point blend; four output pixels per thread, one dword store.
constexpr int kResizeRows = 6;               // output rows per workgroup
// A workgroup produces a 1024-pixel-wide strip of kResizeRows output rows: the source rows they
// sample (kResizeRows * scale + 2 of them) are staged ONCE as aligned dwords (the byte gathers of a
// 1.2x resample would otherwise be 16 scattered loads per thread and row), a thread keeps its four
// columns' table entries in registers for all rows.  One row per workgroup (the first version) was
// bound by workgroup turnaround: 1.4 M short workgroups per 256 pairs, each load -> barrier -> blend.
// The LDS image is sized for THIS level's scale (11 KB at 1.2): in overlap mode the previous batch's
// EPnP workgroups hold 147 of a CU's 160 KB, and a resize workgroup has to fit beside them.
__global__ __launch_bounds__(256) void orb_resize_kernel(OrbGeom g, uint8_t *slots, int64_t slot_stride, int l,
                                                         const int2 *xtab, const int4 *ytab, int src_rows_cap, int kResizeDw, OrbL0 z)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t rs_rows[];       // src_rows_cap x kResizeDw
    const int b = blockIdx.z, tid = threadIdx.x;
    const bool in_place = l == 1 && z.img != nullptr;                 // level 0 = the input image itself
    const int dw = g.w[l], dh = g.h[l], sp = in_place ? z.pitch : g.pitch[l - 1];
    const int dy0 = blockIdx.y * kResizeRows, dy1 = min(dy0 + kResizeRows, dh);
    const int dx_first = blockIdx.x * 1024, dx_last = min(dx_first + 1023, dw - 1);
    uint8_t *slot = slots + (int64_t)b * slot_stride;
    const uint8_t *src = in_place ? orb_level0(z, b) : slot + g.origin[l - 1];
    const int2 *xt = xtab + g.xtab_off[l];
    const int4 *yt = ytab + g.ytab_off[l];
    // source rows [sy_first, sy_last] (the table's rows are monotone), columns [s0, s0 + 4 ndw)
    const int sy_first = yt[dy0].x, sy_last = yt[dy1 - 1].y;
    const int nsrc = sy_last - sy_first + 1;
    // staged as 16-byte chunks (pixel (0,0) of a level and its pitch are 16-byte aligned; kResizeDw is a multiple of 4):
    // four times the bytes in flight per load of the dword version, which left level 1 (whose source comes from HBM)
    // at 1.5 TB/s
    const int s0 = (xt[dx_first].x & 0xFFFF) & ~15;
    const int ndw = ((((xt[dx_last].x >> 16) - s0) >> 4) + 1) << 2;
    const bool staged = ndw <= kResizeDw && nsrc <= src_rows_cap;
    if (staged) {
        const int n16 = ndw >> 2;
        const float inv_n16 = 1.0f / (float)n16;
        for (int i = tid; i < nsrc * n16; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv_n16), c = i - r * n16;
            *(uint4 *)(rs_rows + r * kResizeDw + 4 * c) = *(const uint4 *)(src + (int64_t)(sy_first + r) * sp + s0 + 16 * c);
        }
        __syncthreads();
    }
    const int dx0 = dx_first + tid * 4;
    if (dx0 >= dw) return;
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int2 t = xt[min(dx0 + q, dw - 1)];                    // sx | sx1 << 16, a0 | a1 << 16
        sx[q] = t.x & 0xFFFF; sx1[q] = t.x >> 16; a0[q] = (short)(t.y & 0xFFFF); a1[q] = t.y >> 16;
    }
    for (int dy = dy0; dy < dy1; dy++) {
        const int4 ty = yt[dy];                                     // y0, y1, b0, b1
        uint32_t out = 0;
        auto blend = [&](const auto *R0, const auto *R1, int base) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r0 = R0[sx[q] - base] * a0[q] + R0[sx1[q] - base] * a1[q];
                const int r1 = R1[sx[q] - base] * a0[q] + R1[sx1[q] - base] * a1[q];
                out |= (uint32_t)((((ty.z * (r0 >> 4)) >> 16) + ((ty.w * (r1 >> 4)) >> 16) + 2) >> 2) << (8 * q);
            }
        };
        if (staged)
            blend((const uint8_t *)(rs_rows + (ty.x - sy_first) * kResizeDw), (const uint8_t *)(rs_rows + (ty.y - sy_first) * kResizeDw), s0);
        else
            blend(src + (int64_t)ty.x * sp, src + (int64_t)ty.y * sp, 0);
        uint8_t *d = slot + g.origin[l] + (int64_t)dy * g.pitch[l] + dx0;       // origin and pitch are 4-byte aligned
        if (dx0 + 4 <= dw) *(uint32_t *)d = out;
        else for (int q = 0; dx0 + q < dw; q++) d[q] = (uint8_t)(out >> (8 * q));
    }
}

// The same resize with the source rows STREAMED through registers (round 5; the staged kernel above stays for scale
// factors this one does not cover, OrbGeom::rs_stream).  cv::resize's linear path is two passes -- every source row is
// blended horizontally ONCE into an int row, an output row blends two of those vertically --, and at a scale of 1.2 a
// source row serves 1.67 output rows: the staged kernel redid the horizontal blend for both rows of every output row
// (4 LDS byte reads + 4 multiplies per pixel).  Here a lane owns four output columns and walks down a band of output
// rows: per SOURCE row three aligned dwords (the 8-byte window from its first source column holds every tap of its four
// pixels while sx1[3] - sx[0] <= 7), two v_alignbyte to normalise the alignment, per pixel one v_perm (the two taps as
// 16-bit halves) + one v_dot2 against the table's (a0 | a1 << 16) + the `>> 4` of the vertical pass kept as `& ~15`; the
// rows r - 1 and r stay in registers.  An output row whose second source row is r is then four times
//   ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2  =  (mulhi24(b0 << 12, S0 & ~15) + mulhi24(b1 << 12, S1 & ~15) + 2) >> 2
// (b <= 2^11 and S < 2^20: both factors fit 24 bits, the 48-bit product >> 32 is the same floor), the four bytes
// gathered by three v_perm.  No LDS, no barrier; source rows are requested ahead of their blend.
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__global__ __launch_bounds__(256) void orb_resize_stream_kernel(OrbGeom g, uint8_t *__restrict__ slots, int64_t slot_stride, int l,
                                                                const int2 *__restrict__ xtab, const int4 *__restrict__ ytab, int n_img, int gx, int gy, int band, OrbL0 z)
{
    int b, blk;
    xcd_image_block(blockIdx.x, gx * gy, n_img, b, blk);
    const int by = blk / gx, bx = blk - by * gx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: the row walk below is scalar control)
    const bool in_place = l == 1 && z.img != nullptr;                 // level 0 = the input image itself
    const int dw = g.w[l], dh = g.h[l], sp = in_place ? z.pitch : g.pitch[l - 1], sw = g.w[l - 1];
    const int dy0 = (by * 4 + wave) * band, dy1 = min(dy0 + band, dh);
    if (dy0 >= dh) return;
    const int dx0 = (bx * 64 + lane) * 4;
    uint8_t *slot = slots + (int64_t)b * slot_stride;
    const int2 *xt = xtab + g.xtab_off[l];
    const int4 *yt = ytab + g.ytab_off[l];
    // the lane's four columns (a lane past the row's end takes the last column four times: its loads stay in the row, it stores nothing)
    uint32_t sel[4], wt[4];
    int sx0;
    {
        int2 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = xt[min(dx0 + q, dw - 1)];
        sx0 = t[0].x & 0xFFFF;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p0 = (t[q].x & 0xFFFF) - sx0, p1 = (t[q].x >> 16) - sx0;          // 0..7: bytes of the window from sx0
            sel[q] = (uint32_t)p0 | 0x0c00u | ((uint32_t)p1 << 16) | 0x0c000000u;
            wt[q] = (uint32_t)t[q].y;
        }
    }
    const int off0 = sx0 & 3, last_dw = (sw - 1) >> 2;
    const int c0 = sx0 >> 2, c1 = min(c0 + 1, last_dw), c2 = min(c0 + 2, last_dw);       // (a clamped dword holds no tap of this lane)
    const uint8_t *src = in_place ? orb_level0(z, b) : slot + g.origin[l - 1];
    uint8_t *dst = slot + g.origin[l] + dx0;
    const int dpitch = g.pitch[l];
    const int rs = yt[dy0].x, re = yt[dy1 - 1].y;
    auto request = [&](uint32_t (&d)[3], int r) {
        const uint32_t *row = (const uint32_t *)(src + (int64_t)r * sp);
        d[0] = row[c0]; d[1] = row[c1]; d[2] = row[c2];
    };
    auto hblend = [&](int (&H)[4], const uint32_t (&d)[3]) {
        const uint32_t A = __builtin_amdgcn_alignbyte(d[1], d[0], off0), B = __builtin_amdgcn_alignbyte(d[2], d[1], off0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t taps = __builtin_amdgcn_perm(B, A, sel[q]);
            int r;
            asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(taps), "v"(wt[q]));      // (the VOP3P form: no zeroed accumulator to set up)
            H[q] = r & ~15;
        }
    };
    int dy = dy0;
    int4 ty = yt[dy];
    auto emit = [&](const int (&H0)[4], const int (&H1)[4]) {      // the output rows whose second source row is the one in H1
        const uint32_t B0 = (uint32_t)ty.z << 12, B1 = (uint32_t)ty.w << 12;
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = ((mulhi_u24(B0, (uint32_t)H0[q]) + mulhi_u24(B1, (uint32_t)H1[q])) << 6) + 128u;   // byte 1 = (sum + 2) >> 2
        const uint32_t lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0501u), hi = __builtin_amdgcn_perm(v[3], v[2], 0x0c0c0501u);
        const uint32_t out = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        // origin and pitch are 4-byte aligned; the row's last dword may run up to three bytes into the 32-byte margin every
        // level has on each side (nothing reads it: the blur reflects, cell FAST and this kernel ignore what they stage past a row)
        if (dx0 < dw) *(uint32_t *)(dst + (int64_t)dy * dpitch) = out;
    };
    int HA[4] = {0, 0, 0, 0}, HB[4] = {0, 0, 0, 0};
    // The next source row is requested before this one is blended; two rows a turn, so that the row buffers and the roles
    // (previous, current: HA / HB) alternate without moves.  (Three rows in flight measured no faster -- eight waves per SIMD
    // hide the round trip, 0.52 against 0.52-0.55 ms per 514 images -- and cost six registers: at 56 instead of 60 a third wave
    // of this kernel fits beside two hypothesis blocks of the previous batch's pose stage.)
    uint32_t dA[3], dB[3];
    request(dA, rs);
#define SVO_RS_STEP(DN, DC, RR, HP, HC)                                                               \
        if ((RR) + 1 <= re) request(DN, (RR) + 1);                                                    \
        hblend(HC, DC);                                                                               \
        while (dy < dy1 && ty.y == (RR)) {                                                            \
            if (__builtin_expect(ty.x == (RR), 0)) emit(HC, HC); else emit(HP, HC);                   \
            if (++dy < dy1) ty = yt[dy];                                                              \
        }
    for (int r = rs; r <= re; r += 2) {
        SVO_RS_STEP(dB, dA, r, HB, HA)
        if (r + 1 > re) break;
        SVO_RS_STEP(dA, dB, r + 1, HA, HB)
    }
#undef SVO_RS_STEP
}

// ---- per-cell FAST ---------------------------------------------------------------------------
constexpr int kCellMax = 66;                 // wCell = ceil(width / floor(width / 30)) < 60, +6 overlap
constexpr int kCellGroup = 4;                // cells of a cell row per workgroup, at most
constexpr int kCellPitch = 136;              // LDS row pitch of a workgroup's window: up to 4 cells of 31 + 6 overlap + 3 bytes of dword-alignment slack, multiple of 4
constexpr int kCellCap = 256;                // candidates kept per cell
// threads per workgroup (measured with one cell per workgroup: 64 -> 3.53 ms per 256 pairs, 128 -> 2.86, 256 -> 2.76: the
// loops' nearly empty last passes cost less than the occupancy smaller workgroups lose)
constexpr int kCellThreads = 256;
static_assert(kCellPitch * kCellMax < (1 << 14), "a window position and a 2-bit cell number share 16 bits");
// The workgroup's dynamic LDS, carved per level: pixel plane + cornerness plane (kCellPitch x (hCell + 6) bytes each), the
// survivor / candidate list (one 16-bit entry per TESTED pixel: the window without its 3-pixel rim), the keypoint list and one
// scratch entry.  Round 6: the lists were three planes' worth (31.3 KB for KITTI's tallest cells: five workgroups per CU); sized
// by what they can hold, 26.4 KB: six.
__host__ __device__ constexpr int cellfast_plane(int hCell) { return (kCellPitch * (hCell + 6) + 15) & ~15; }
__host__ __device__ constexpr int cellfast_listcap(int hCell) { return hCell * (kCellPitch - 6); }
__host__ __device__ constexpr int cellfast_klistcap(int hCell) { return (kCellPitch / 2 + 4) * ((hCell + 6) / 2 + 1); }
__host__ __device__ constexpr size_t cellfast_lds_bytes(int hCell)
{
    return (size_t)2 * cellfast_plane(hCell) + 2 * (size_t)(cellfast_listcap(hCell) + cellfast_klistcap(hCell) + 1) + 14;
}

// FAST cornerness V = largest t for which the pixel is a FAST-9/16 corner (0 when < 1):
// corner at threshold t <=> V >= t, and cornerScore == V.
__device__ __forceinline__ int fast_cornerness(const uint8_t *c, int P)
{
    const int v = c[0];
    int d[16];
    d[0] = c[3 * P];       d[1] = c[3 * P + 1];   d[2] = c[2 * P + 2];   d[3] = c[P + 3];
    d[4] = c[3];           d[5] = c[-P + 3];      d[6] = c[-2 * P + 2];  d[7] = c[-3 * P + 1];
    d[8] = c[-3 * P];      d[9] = c[-3 * P - 1];  d[10] = c[-2 * P - 2]; d[11] = c[-P - 3];
    d[12] = c[-3];         d[13] = c[P - 3];      d[14] = c[2 * P - 2];  d[15] = c[3 * P - 1];
    // The cornerness is max over the 9-arcs of min(v - p) and of min(p - v): on the ring values themselves that is
    // v - (smallest arc maximum) and (largest arc minimum) - v -- no sixteen differences.  Min / max over every 9-arc
    // d[i..i+8] as three-input ops: triples, then triples of triples.
    int m3[16], x3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        m3[i] = min(min(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
        x3[i] = max(max(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
    }
    int max_of_min = 0, min_of_max = 255;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        max_of_min = max(max_of_min, min(min(m3[i], m3[(i + 3) & 15]), m3[(i + 6) & 15]));
        min_of_max = min(min_of_max, max(max(x3[i], x3[(i + 3) & 15]), x3[(i + 6) & 15]));
    }
    const int V = max(v - min_of_max, max_of_min - v) - 1;
    return V > 0 ? V : 0;
}

// One workgroup per GROUP of up to four consecutive cells of a cell row; one-dimensional grid of blks_total x n_img
// blocks, image by XCD (xcd_image_block).  The reference runs FAST on every cell's own window (the cell + 3 pixels on
// each side), so neighbouring windows overlap by 6 pixels and every window is a separate little image: its corners lie
// 3 pixels inside it and its non-maximum suppression sees nothing outside those.  The windows of a group are taken as ONE
// window -- the test and the cornerness of a pixel do not depend on the window it is seen from -- in which a pixel
// belongs to the cell whose interior holds it, and the suppression ignores neighbours that belong to another cell.
// Per cell the fixed work (staging, barriers, list bookkeeping, nearly empty last passes of the loops) is a quarter
// of the one-cell-per-workgroup kernel's, and the overlap columns are tested once instead of twice.
// Diagnostic build (-DSVO_CF_STAMP, tools/gpu/cf_stamps.sh): a few workgroups of image 2 print the s_memtime ticks their first
// wave spent per section.
#ifdef SVO_CF_STAMP
#define CF_AT(i) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime()); cf_acc[i] += now_ - cf_last; cf_last = now_; }
#else
#define CF_AT(i)
#endif
__global__ __launch_bounds__(kCellThreads) void orb_cellfast_kernel(OrbGeom g, const uint8_t *slots, int64_t slot_stride,
                                                           int iniTh, int minTh, float4 *cell_cand, int *cell_cnt,
                                                           int64_t cand_img_stride, int64_t cnt_img_stride, int n_img, OrbL0 z,
                                                           int blk_lo, int blk_n)       // this launch: workgroups blk_lo .. blk_lo + blk_n - 1 of every image
{
#ifdef SVO_CF_STAMP
    uint32_t cf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cf_last = 0;
    CF_AT(7) cf_acc[7] = 0;
#endif
    int b, blk_all;                                                       // image, workgroup index over all levels
    xcd_image_block(blockIdx.x, blk_n, n_img, b, blk_all);
    blk_all += blk_lo;
    const int l = level_of_block(g.blk_off, g.nlevels, blk_all);
    // dynamic LDS: two byte planes (pixels, cornerness) + the position lists (three planes' worth), kCellPitch
    // columns x (hCell + 6) rows of THIS level
    extern __shared__ __attribute__((aligned(16))) uint8_t cf_smem[];
    const int plane = cellfast_plane(g.hCell[l]);
    uint8_t *raw = cf_smem, *V = cf_smem + plane;
    uint16_t *list = (uint16_t *)(cf_smem + 2 * plane);      // survivors of the test, then candidates (compacted in place): every TESTED pixel can survive
    uint16_t *klist = list + cellfast_listcap(g.hCell[l]);   // keypoints after the NMS, in any order: position | cell << 14 (strict 3 x 3 suppression
                                                             // inside a cell: at most every other pixel of every other row of each cell)
    __shared__ int s_any[kCellGroup], s_nkept[kCellGroup], s_nlist, s_ncand, s_nk;
    const int blk = blk_all - g.blk_off[l];
    const bool in_place = l == 0 && z.img != nullptr;                     // level 0 = the input image itself
    const int W = g.w[l], H = g.h[l], pitch = in_place ? z.pitch : g.pitch[l];
    const int minBX = 16, minBY = 16, maxBX = W - 16, maxBY = H - 16;
    const int nCols = g.nCols[l], wCell = g.wCell[l], hCell = g.hCell[l], gcols = g.gcols[l], G = g.gcell[l];
    const int ci = blk / gcols, cg = blk - ci * gcols, cj0 = cg * G;
    const int ngroup = min(G, nCols - cj0);                               // cells of this group
    const float iniY = (float)(minBY + ci * hCell), iniX = (float)(minBX + cj0 * wCell);
    float maxY = iniY + hCell + 6;
    int *cnt = cell_cnt + (int64_t)b * cnt_img_stride + g.cell_off[l] + ci * nCols + cj0;
    const int tid = threadIdx.x;
    if (maxY > maxBY) maxY = (float)maxBY;
    const int x0 = (int)iniX, y0 = (int)iniY, ch = (int)maxY - y0;
    // the cells that take part: not skipped by the reference (iniX >= maxBorderX - 6) and with a window of at least 7
    // columns (FAST finds nothing in a narrower image); they are the first ncl of the group.  cw = width of their common window.
    int ncl = 0, cw = 0;
    for (int c = 0; c < ngroup; c++) {
        const int ix = x0 + c * wCell;
        const int wc = min(ix + wCell + 6, maxBX) - ix;
        if (ix >= maxBX - 6 || wc < 7) break;
        ncl = c + 1; cw = c * wCell + wc;
    }
    if (iniY >= maxBY - 3 || ch < 7 || ncl == 0) { if (tid < ngroup) cnt[tid] = 0; return; }
    // (row, column) of a linear index without an integer division (a runtime divisor costs ~25 vector
    // instructions): (i + 0.5) / n is never within 1e-3 of an integer for the i, n that occur here,
    // far above the float error, so truncating (i + 0.5) * (1 / n) is exact
    const int lowTh = min(iniTh, minTh);
    const uint8_t *img = in_place ? orb_level0(z, b) : slots + (int64_t)b * slot_stride + g.origin[l];
    if (tid < kCellGroup) { s_any[tid] = 0; s_nkept[tid] = 0; }
    if (tid == 0) { s_nlist = 0; s_ncand = 0; s_nk = 0; }
    // V must read 0 wherever the NMS looks and no cornerness is computed.  The rows of the tested pixels are
    // zeroed by the test loop itself (one dword store beside its five reads); the row above and the row
    // below them here (a whole-plane clear cost five passes of the workgroup)
    uint32_t *Vd = (uint32_t *)V;                            // aligned dword view, like rawd
    constexpr int RD = kCellPitch / 4;
    if (tid < 2 * RD) Vd[(tid < RD ? 2 : ch - 3) * RD + tid % RD] = 0;
    // the window as aligned dwords (rows of the level are 4-byte aligned; the window's first column
    // sits `ox` bytes into its first dword, so every LDS row is shifted by ox: rawc = raw + ox)
    const int ox = x0 & 3, nd = (ox + cw + 3) >> 2;
    const float inv_nd = 1.0f / (float)nd;
    {
        const uint8_t *src = img + (int64_t)y0 * pitch + (x0 - ox);
        for (int i = tid; i < nd * ch; i += kCellThreads) {
            const int y = (int)(((float)i + 0.5f) * inv_nd), c = i - y * nd;
            ((uint32_t *)(raw + y * kCellPitch))[c] = *(const uint32_t *)(src + (int64_t)y * pitch + 4 * c);
        }
    }
    const uint32_t *rawd = (const uint32_t *)raw;           // aligned dword view, RD dwords per row
    raw += ox;
    V += ox;                                                 // the same shift: position p of the window is raw[p] and V[p]
    // the cell of a window column (its interior is [3 + c wCell, 3 + (c + 1) wCell), the last one ends at cw - 3)
    auto cell_of = [&](int x) { const int xi = x - 3; return (xi >= wCell) + (xi >= 2 * wCell) + (xi >= 3 * wCell); };
    __syncthreads();
    CF_AT(0)                                                 // prologue + staging
    // Cornerness only matters where it can reach the threshold in force: a corner at threshold t needs one
    // pixel of each opposite pair (0,8), (4,12) beyond t, so positions failing that 4-pixel test at t keep
    // V = 0, and the full arc min/max runs over a compacted list of the survivors.
    // The reference runs FAST at iniTh and, only if the cell stays empty, again at minTh.  Cornerness is
    // therefore computed in TWO PHASES: first for the survivors of the test at iniTh (on textured images
    // nearly every pixel survives the test at minTh = 7, and the arc min/max of every pixel was three
    // quarters of this kernel); the pixels with iniTh > V >= minTh only matter when a cell has no
    // keypoint at iniTh (no corner, or strict NMS emptied it), and only then are they computed.
    // The test runs on FOUR pixels per thread (one aligned LDS dword of row y and its neighbours
    // three rows up / down and three columns left / right: five dword reads instead of twenty byte
    // reads), in packed 16-bit arithmetic on the even and the odd bytes:
    //   alive <=> max( min(v - min(p0,p8), v - min(p4,p12)),  min(max(p0,p8) - v, max(p4,p12) - v) ) > t
    // the test loop's thread -> (column group, row) map: nd column groups x tR rows per step
    const int tR = kCellThreads / nd;
    const int tRow = (int)(((float)tid + 0.5f) * inv_nd), tq = tid - tRow * nd;
    const int tXb = 4 * tq - ox;                                             // window x of the column group's first byte
    uint32_t tInside;                                                        // its pixels inside [3, cw - 3)
    {
        const int lo = max(0, 3 - tXb), hi = min(4, cw - 3 - tXb);
        tInside = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    }
    auto survivors_and_cornerness = [&](const int th) {
        {
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const int rows = ch - 6;
            const s16x2 T1 = {(short)(th + 1), (short)(th + 1)};
            // a thread keeps ONE column group (four pixels) and walks down the rows, R rows of the window per step: the
            // column's position, its in-window mask and its left neighbour are set up once
            const int lane = tid & 63;
            // the survivors of a thread's column group: four bits per step, compacted ONCE after the walk (a prefix sum and an
            // LDS atomic per step were 40 % of this loop)
            unsigned long long alive = 0;
            int step = 0;
            for (int yb = 0; yb < rows; yb += tR, step++) {
                const int yy = yb + tRow;
                const int y = yy + 3;
                if (tRow < tR && yy < rows) {
                    const uint32_t *r = rawd + y * RD + tq;
                    const uint32_t C = r[0], U = r[-3 * RD], D = r[3 * RD], Lf = tq > 0 ? r[-1] : 0u, Rt = r[1];
                    Vd[y * RD + tq] = 0;                                     // (see the clear of the two outer rows above)
                    const uint32_t Lv = __builtin_amdgcn_alignbyte(C, Lf, 1);    // x - 3 neighbours of the four pixels
                    const uint32_t Rv = __builtin_amdgcn_alignbyte(Rt, C, 3);    // x + 3 neighbours
                    uint32_t sgn[2];
    #pragma unroll
                    for (int hb = 0; hb < 2; hb++) {                          // even bytes (pixels 0, 2), odd bytes (1, 3)
                        auto half = [&](uint32_t w) { return __builtin_bit_cast(s16x2, (hb ? w >> 8 : w) & 0x00FF00FFu); };
                        const s16x2 v = half(C), u = half(U), d = half(D), l = half(Lv), rr = half(Rv);
                        const s16x2 mnA = __builtin_elementwise_min(u, d), mxA = __builtin_elementwise_max(u, d);
                        const s16x2 mnB = __builtin_elementwise_min(l, rr), mxB = __builtin_elementwise_max(l, rr);
                        const s16x2 dark = v - __builtin_elementwise_max(mnA, mnB);      // == min(v - mnA, v - mnB)
                        const s16x2 bright = __builtin_elementwise_min(mxA, mxB) - v;    // == min(mxA - v, mxB - v)
                        sgn[hb] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(dark, bright) - T1);   // sign set <=> not alive
                    }
                    const uint32_t dead = ((sgn[0] >> 15) & 1u) | ((sgn[1] >> 14) & 2u) | ((sgn[0] >> 29) & 4u) | ((sgn[1] >> 28) & 8u);
                    alive |= (unsigned long long)(~dead & tInside) << (4 * step);
                }
            }
            // survivors -> list (any order): a prefix sum of the lanes' counts and one LDS atomic per wave; a lane's entries
            // are written without branches (a dead pixel's store goes to a scratch entry past the keypoint list)
            const int mine = __popcll(alive);
            const int incl = wave_incl_scan(mine);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total) {
                int base = 0;
                if (lane == 63) base = atomicAdd(&s_nlist, total);
                int at = __builtin_amdgcn_readlane(base, 63) + incl - mine;
                uint16_t *scratch = klist + cellfast_klistcap(g.hCell[l]);
                for (int st = 0; st < step; st++) {
                    const uint32_t m4 = (uint32_t)(alive >> (4 * st)) & 15u;
                    const int pos = (3 + tRow + st * tR) * kCellPitch + tXb;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool on = (m4 >> q) & 1u;
                        *(on ? list + at : scratch) = (uint16_t)(pos + q);
                        at += on;
                    }
                }
            }
        }
        __syncthreads();
        CF_AT(1)                                             // 4-pixel test walk + survivor list
        // cornerness of the survivors; the positions that reach the threshold are compacted again IN PLACE (a
        // write index never passes the block of kCellThreads entries being read), so the NMS passes below only
        // visit possible keypoints instead of every pixel of the window
        const int nlist = s_nlist;
        for (int i0 = 0; i0 < nlist; i0 += kCellThreads) {
            const int i = i0 + tid;
            int pos = 0, v = 0;
            if (i < nlist) {
                pos = list[i];
                v = fast_cornerness(&raw[pos], kCellPitch);
                V[pos] = (uint8_t)v;
                if (v >= iniTh) {                                            // "FAST at iniTh finds something in this cell"
                    const int y = (int)(((float)pos + 0.5f) * (1.0f / (float)kCellPitch));
                    s_any[cell_of(pos - y * kCellPitch)] = 1;
                }
            }
            __syncthreads();
            const bool cand = v >= th && v > 0;
            const unsigned long long m = __ballot(cand);
            if (m) {
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_ncand, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (cand) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)pos;
            }
        }
        CF_AT(2)                                             // cornerness of the survivors + candidate list
    };
    // phase 1 at the higher of the two thresholds a cell can end up with, phase 2 (all pixels that can
    // reach the lower one) only when the first NMS pass leaves a cell empty
    const int th1 = iniTh > minTh ? iniTh : lowTh;
    int th_done = th1;
    survivors_and_cornerness(th1);
    __syncthreads();
    int ncand = s_ncand;
    int thr[kCellGroup];
    bool open[kCellGroup];                                   // cells still without keypoints
#pragma unroll
    for (int c = 0; c < kCellGroup; c++) { thr[c] = s_any[c] ? iniTh : minTh; open[c] = c < ncl; }
    for (int pass = 0; pass < 2; pass++) {
        int low = 0x7FFFFFFF;
#pragma unroll
        for (int c = 0; c < kCellGroup; c++) if (open[c]) low = min(low, thr[c]);
        if (low < th_done) {
            // the pixels with th_done > V >= minTh are needed now: V of every survivor of the weaker test
            // (those of phase 1 come out the same again), candidate list rebuilt
            __syncthreads();
            if (tid == 0) { s_nlist = 0; s_ncand = 0; }
            __syncthreads();
            survivors_and_cornerness(lowTh);
            th_done = lowTh;
            __syncthreads();
            ncand = s_ncand;
        }
        // strict 3x3 NMS over the candidates of the open cells, among the pixels of the SAME cell; the keypoints go to
        // klist through one LDS atomic per wave and pass
        for (int i0 = 0; i0 < ncand; i0 += kCellThreads) {
            const int i = i0 + tid;
            int pos = 0, c = 0;
            bool k = false;
            if (i < ncand) {
                pos = list[i];
                const uint8_t *p = &V[pos];
                const int s = p[0];
                const int y = (int)(((float)pos + 0.5f) * (1.0f / (float)kCellPitch)), x = pos - y * kCellPitch;
                c = cell_of(x);
                const int t = c == 0 ? thr[0] : c == 1 ? thr[1] : c == 2 ? thr[2] : thr[3];
                const bool op = c == 0 ? open[0] : c == 1 ? open[1] : c == 2 ? open[2] : open[3];
                if (op && s >= t) {
                    const bool lok = x - 3 != c * wCell, rok = x - 3 != (c + 1) * wCell - 1;     // neighbours in the same cell
#define SC(o) (p[o] >= t ? (int)p[o] : 0)
                    k = s > SC(-kCellPitch) && s > SC(kCellPitch) &&
                        (!lok || (s > SC(-1) && s > SC(-kCellPitch - 1) && s > SC(kCellPitch - 1))) &&
                        (!rok || (s > SC(1) && s > SC(-kCellPitch + 1) && s > SC(kCellPitch + 1)));
#undef SC
                }
            }
            const unsigned long long m = __ballot(k);
            if (m) {
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nk, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (k) { klist[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(pos | (c << 14)); atomicAdd(&s_nkept[c], 1); }
            }
        }
        __syncthreads();
        CF_AT(3)                                             // NMS pass
        // "if (vKeysCell.empty()) FAST(..., minThFAST)": strict NMS can empty a cell whose corners tie
        bool again = false;
#pragma unroll
        for (int c = 0; c < kCellGroup; c++) {
            if (!open[c]) continue;
            if (s_nkept[c] > 0 || thr[c] == minTh) open[c] = false;
            else { thr[c] = minTh; again = true; }
        }
        if (!again) break;
    }
    // ordered (row-major) emission: a cell keeps a handful of keypoints (entries cell << 14 | y * pitch + x, so their
    // numeric order IS the cell order, then the row-major order), and a keypoint's slot is the number of keypoints with a
    // smaller entry minus those of the cells before its own -- counted by its thread over the list (LDS broadcast reads)
    {
        const int nk = s_nk;
        int nkc[kCellGroup], before[kCellGroup];
#pragma unroll
        for (int c = 0, run = 0; c < kCellGroup; c++) { nkc[c] = s_nkept[c]; before[c] = run; run += nkc[c]; }
        for (int i = tid; i < nk; i += kCellThreads) {
            const int e = klist[i], c = e >> 14, pos = e & 0x3FFF;
            int rank = 0;
            for (int j = 0; j < nk; j++) rank += klist[j] < e;
            rank -= c == 0 ? before[0] : c == 1 ? before[1] : c == 2 ? before[2] : before[3];
            if (rank < kCellCap) {
                const int y = (int)(((float)pos + 0.5f) * (1.0f / (float)kCellPitch)), x = pos - y * kCellPitch;
                float4 *out = cell_cand + (int64_t)b * cand_img_stride + ((int64_t)g.cell_off[l] + ci * nCols + cj0 + c) * kCellCap;
                out[rank] = make_float4((float)(x - c * wCell) + (float)((cj0 + c) * wCell), (float)y + (float)(ci * hCell), (float)V[pos], 0.f);
            }
        }
        // (a count may exceed kCellCap: flagged by orb_gather_kernel)
        if (tid < ngroup) cnt[tid] = tid == 0 ? nkc[0] : tid == 1 ? nkc[1] : tid == 2 ? nkc[2] : nkc[3];
        CF_AT(4)                                             // ordered emission
#ifdef SVO_CF_STAMP
        if (tid == 0 && b == 2 && (blk == 0 || blk == 37) && (l == 0 || l == 2 || l == 5))
            printf("cf l %d blk %d cw %d ch %d nlist %d ncand %d nk %d | stage %u test %u corner %u nms %u emit %u\n", l, blk, cw, ch, s_nlist, s_ncand, nk,
                   cf_acc[0], cf_acc[1], cf_acc[2], cf_acc[3], cf_acc[4]);
#endif
    }
}

// cells -> level candidate list (cells row-major): one workgroup per (level, image)
__global__ __launch_bounds__(256) void orb_gather_kernel(OrbGeom g, const float4 *cell_cand, const int *cell_cnt,
                                                         int64_t cand_img_stride, int64_t cnt_img_stride,
                                                         float4 *lvl_cand, int *lvl_cnt, int cand_cap, int *overflow)
{
    __shared__ int offs[1024];
    __shared__ unsigned short cn[1024];
    __shared__ int total;
    const int l = blockIdx.x, b = blockIdx.y;
    const int ncell = g.ncell[l];
    const int *cnt = cell_cnt + (int64_t)b * cnt_img_stride + g.cell_off[l];
    // exclusive prefix of the per-cell counts: four cells per thread, wave scan, scan of the four wave totals
    {
        __shared__ int wave_sum[4];
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        int v[4], mine = 0;
        bool ovf = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid * 4 + q;
            const int n = c < ncell ? cnt[c] : 0;
            ovf = ovf || n > kCellCap;
            v[q] = min(n, kCellCap);
            mine += v[q];
        }
        int incl = mine;                                   // inclusive scan across the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_sum[wv] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wv; w++) base += wave_sum[w];
        int run = base + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c = tid * 4 + q;
            if (c < ncell) { offs[c] = run; cn[c] = (unsigned short)v[q]; }
            run += v[q];
        }
        if (tid == 255) total = run;                       // ncell <= 1024 == 4 * 256: the last thread ends the scan
        const unsigned long long any_ovf = __ballot(ovf);
        if (lane == 0 && any_ovf) atomicOr(overflow + b, 2);
    }
    __syncthreads();
    if (threadIdx.x == 0 && total > cand_cap) atomicOr(overflow + b, 2);
    const float4 *src = cell_cand + (int64_t)b * cand_img_stride + (int64_t)g.cell_off[l] * kCellCap;
    float4 *dst = lvl_cand + ((int64_t)b * g.nlevels + l) * cand_cap;
    // a cell holds a handful of candidates: 8 lanes per cell, 32 cells per pass, counts and offsets from the LDS (with 16
    // lanes and the count read from global memory a level-0 list was 28 passes of two dependent loads: the launch's time)
    for (int c = threadIdx.x >> 3; c < ncell; c += 32) {
        const int n = cn[c], o = offs[c];
        for (int k = threadIdx.x & 7; k < n; k += 8)
            if (o + k < cand_cap) dst[o + k] = src[(int64_t)c * kCellCap + k];
    }
    if (threadIdx.x == 0) lvl_cnt[b * g.nlevels + l] = min(total, cand_cap);
}

// ---- DistributeOctTree: one wavefront per (level, image) ----------------------------------------
// The tree, its std::list order and the candidate keys live in LDS.  Control flow is wave-uniform
// (the list walk, largest-first expansion and termination rules of ORBextractor.cpp:487-715 are
// serial by definition), while everything that touches keys is done by the 64 lanes together:
// DivideNode is a stable in-place 4-way partition of the node's key range (ballot ranks), the
// "sort by size" of the last phase is a rank sort, and the final best-response pick runs one leaf
// per lane.  A key is x | y << 16 (cell-space pixel coordinates) plus a 16-bit candidate index (two
// planes, 6 bytes per key; responses are read from the candidate list at the end); the partition is
// stable, so a node's keys stay in candidate order and "first maximum" == the first key with the
// largest response.
constexpr int kLdsKeys = 3456;    // candidates partitioned in LDS (39 KB per instance at the default quota: four per CU, one per SIMD); larger inputs use global scratch

struct QBox { short ulx, uly, brx, bry; };
// dynamic LDS of one instance, carved for `node_cap` live nodes (<= quota + 3 leaves, + 4 children in
// flight; the host sizes it from the largest per-level quota of the configuration)
// A key is (x | y << 16 in cell-space pixels, candidate index): two planes, 6 bytes per key.
struct KeyArr {
    uint32_t *xy; uint16_t *id;
    __device__ __forceinline__ uint2 get(int i) const { return make_uint2(xy[i], id[i]); }
    __device__ __forceinline__ void set(int i, uint2 v) const { xy[i] = v.x; id[i] = (uint16_t)v.y; }
};
struct QLds {
    KeyArr keys;
    unsigned long long *ea, *eb;          // (count << 40 | seq << 12 | id) of nodes still to expand
    QBox *box;
    unsigned short *begin, *count, *seq;
    short *prev, *next, *free_list;
    int node_cap;
};
__host__ __device__ inline size_t qlds_bytes(int node_cap)
{
    return (size_t)kLdsKeys * 6 + (size_t)node_cap * (8 + 8 + 8 + 2 * 3 + 2 * 3) + 16;
}
__device__ inline QLds qlds_carve(uint8_t *smem, int node_cap)
{
    QLds L;
    L.node_cap = node_cap;
    L.keys.xy = (uint32_t *)smem; smem += (size_t)kLdsKeys * 4;
    L.keys.id = (uint16_t *)smem; smem += (size_t)kLdsKeys * 2;
    L.ea = (unsigned long long *)smem; smem += (size_t)node_cap * 8;
    L.eb = (unsigned long long *)smem; smem += (size_t)node_cap * 8;
    L.box = (QBox *)smem; smem += (size_t)node_cap * 8;
    L.begin = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.count = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.seq = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.prev = (short *)smem; smem += (size_t)node_cap * 2;
    L.next = (short *)smem; smem += (size_t)node_cap * 2;
    L.free_list = (short *)smem;
    return L;
}
struct QState { int n_free, n_alloc, head, tail, size, seq; bool overflow; };

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline int qt_new(QLds &L, QState &t, int lane)
{
    int id;
    if (t.n_free > 0) id = rfl(L.free_list[--t.n_free]);
    else if (t.n_alloc < L.node_cap) id = t.n_alloc++;
    else { t.overflow = true; id = L.node_cap - 1; }
    if (lane == 0) L.seq[id] = (unsigned short)t.seq;
    t.seq++;
    return id;
}
__device__ inline void qt_push_front(QLds &L, QState &t, int id, int lane)
{
    if (lane == 0) { L.next[id] = (short)t.head; L.prev[id] = -1; if (t.head >= 0) L.prev[t.head] = (short)id; }
    if (t.head < 0) t.tail = id;
    t.head = id; t.size++;
}
__device__ inline void qt_push_back(QLds &L, QState &t, int id, int lane)
{
    if (lane == 0) { L.prev[id] = (short)t.tail; L.next[id] = -1; if (t.tail >= 0) L.next[t.tail] = (short)id; }
    if (t.tail < 0) t.head = id;
    t.tail = id; t.size++;
}
__device__ inline void qt_erase(QLds &L, QState &t, int id, int lane)
{
    const int p = rfl(L.prev[id]), n = rfl(L.next[id]);
    if (lane == 0) {
        if (p >= 0) L.next[p] = (short)n;
        if (n >= 0) L.prev[n] = (short)p;
        L.free_list[t.n_free] = (short)id;
    }
    if (p < 0) t.head = n;
    if (n < 0) t.tail = p;
    t.size--; t.n_free++;
}

// All scalar fields of a node in one batch of LDS reads (one round trip instead of one per field:
// the tree walk is a chain of dependent LDS accesses, and its latency is what the kernel costs)
struct NodeView { int ulx, uly, brx, bry, beg, cnt, prev, next; };
__device__ __forceinline__ NodeView qt_view(const QLds &L, int id)
{
    const QBox P = L.box[id];
    const int b = L.begin[id], c = L.count[id], p = L.prev[id], n = L.next[id];
    NodeView v;
    v.ulx = rfl(P.ulx); v.uly = rfl(P.uly); v.brx = rfl(P.brx); v.bry = rfl(P.bry);
    v.beg = rfl(b); v.cnt = rfl(c); v.prev = rfl(p); v.next = rfl(n);
    return v;
}
// erase with the neighbours already known (no reads)
__device__ inline void qt_erase_known(QLds &L, QState &t, int id, int p, int n, int lane)
{
    if (lane == 0) {
        if (p >= 0) L.next[p] = (short)n;
        if (n >= 0) L.prev[n] = (short)p;
        L.free_list[t.n_free] = (short)id;
    }
    if (p < 0) t.head = n;
    if (n < 0) t.tail = p;
    t.size--; t.n_free++;
}

__device__ __forceinline__ int key_quadrant(uint2 k, int midx, int midy)
{
    return ((int)(k.x & 0xFFFFu) < midx ? 0 : 1) + ((int)(k.x >> 16) < midy ? 0 : 2);
}
// rank of this lane's key inside its quadrant for one 64-key chunk; adds the chunk's counts to c[]
__device__ __forceinline__ int chunk_rank(bool valid, int q, int lane, int c[4])
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    int r = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long m = __ballot(valid && q == k);
        if (q == k) r = c[k] + __popcll(m & lt);
        c[k] += __popcll(m);
    }
    return r;
}

// ExtractorNode::DivideNode (ORBextractor.cpp:430-485): stable 4-way partition of the node's keys,
// children created for the non-empty quadrants (ch[q] = node id or -1, cnt[q] = its key count)
__device__ inline void qt_partition(const KeyArr keys, const KeyArr gtmp, bool keys_global, const NodeView &nv, int midx, int midy,
                                    int cnt[4], int lane)
{
    const int beg = nv.beg, n = nv.cnt;
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    if (n <= 64) {
        const bool valid = lane < n;
        const uint2 k = valid ? keys.get(beg + lane) : make_uint2(0, 0);
        const int q = key_quadrant(k, midx, midy);
        const int r = chunk_rank(valid, q, lane, cnt);
        const int off = q == 0 ? 0 : (q == 1 ? cnt[0] : (q == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2]));
        if (valid) keys.set(beg + off + r, k);
    } else if (n <= 512) {
        uint2 k[8];
        int q[8], r[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            k[c] = make_uint2(0, 0); q[c] = 0; r[c] = 0;
            if (64 * c >= n) continue;                        // (uniform: chunks past the node's last key cost nothing)
            const bool valid = lane + 64 * c < n;
            k[c] = valid ? keys.get(beg + lane + 64 * c) : make_uint2(0, 0);
            q[c] = key_quadrant(k[c], midx, midy);
            r[c] = chunk_rank(valid, q[c], lane, cnt);
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (64 * c >= n) continue;
            const int off = q[c] == 0 ? 0 : (q[c] == 1 ? cnt[0] : (q[c] == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2]));
            if (lane + 64 * c < n) keys.set(beg + off + r[c], k[c]);
        }
    } else {
        for (int base = 0; base < n; base += 64) {
            const bool valid = base + lane < n;
            const uint2 k = valid ? keys.get(beg + base + lane) : make_uint2(0, 0);
            chunk_rank(valid, key_quadrant(k, midx, midy), lane, cnt);
        }
        int run[4] = {0, 0, 0, 0};
        const int o1 = cnt[0], o2 = cnt[0] + cnt[1], o3 = cnt[0] + cnt[1] + cnt[2];
        for (int base = 0; base < n; base += 64) {
            const bool valid = base + lane < n;
            const uint2 k = valid ? keys.get(beg + base + lane) : make_uint2(0, 0);
            const int q = key_quadrant(k, midx, midy);
            const int r = chunk_rank(valid, q, lane, run);
            const int off = q == 0 ? 0 : (q == 1 ? o1 : (q == 2 ? o2 : o3));
            if (valid) gtmp.set(off + r, k);
        }
        __threadfence();
        for (int i = lane; i < n; i += 64) keys.set(beg + i, gtmp.get(i));
    }
    if (keys_global) __threadfence();
}

__device__ __forceinline__ unsigned long long exp_key(int count, int seq, int id)
{
    return ((unsigned long long)count << 40) | ((unsigned long long)seq << 12) | (unsigned long long)id;
}

// DivideNode + the list surgery around it (ORBextractor.cpp:430-485 and :573-618 / :640-681): the node's keys are
// partitioned, its non-empty children are created and pushed to the FRONT of the list in quadrant order, the
// children with more than one key are appended to the expansion list, the node itself is erased.
// Children are handled by lanes 0..3 AT ONCE (lane = quadrant): ids, creation numbers, boxes, key ranges and the
// list links of all four are written by one vector store each, where the scalar version walked the four quadrants
// one after the other with ~25 single-lane LDS stores and the bookkeeping between them (the tree walk is a
// chain of dependent instructions on a lone wave: their number is what the kernel costs).
// The resulting state is exactly the sequential one: child of rank r (among the non-empty quadrants) takes the
// r-th id off the free list (then fresh ids), creation number seq + r; the list reads
//   last child, ..., first child, <what followed: the old head, or the erased node's successor if it WAS the head>
__device__ inline void qt_split(QLds &L, QState &t, const KeyArr keys, const KeyArr gtmp, bool keys_global, const NodeView &nv,
                                int node, int lane, int &n_exp, int &n_to_expand)
{
    const int ulx = nv.ulx, uly = nv.uly, brx = nv.brx, bry = nv.bry, beg = nv.beg;
    const int midx = ulx + ((brx - ulx + 1) >> 1), midy = uly + ((bry - uly + 1) >> 1);   // ceil(d / 2)
    int cnt[4];
    qt_partition(keys, gtmp, keys_global, nv, midx, midy, cnt, lane);
    // ---- uniform bookkeeping
    const uint32_t ne_mask = (cnt[0] > 0 ? 1u : 0u) | (cnt[1] > 0 ? 2u : 0u) | (cnt[2] > 0 ? 4u : 0u) | (cnt[3] > 0 ? 8u : 0u);
    const uint32_t ex_mask = (cnt[0] > 1 ? 1u : 0u) | (cnt[1] > 1 ? 2u : 0u) | (cnt[2] > 1 ? 4u : 0u) | (cnt[3] > 1 ? 8u : 0u);
    const int k = __popc(ne_mask), g = __popc(ex_mask);
    const int nf0 = t.n_free, take = k < nf0 ? k : nf0, fresh = k - take;
    const int head0 = t.head, seq0 = t.seq, alloc0 = t.n_alloc;
    if (alloc0 + fresh > L.node_cap) t.overflow = true;
    // ---- per quadrant (lanes 0..3)
    const int q = lane & 3;
    const bool mine = lane < 4 && ((ne_mask >> q) & 1u);
    const int r = __popc(ne_mask & ((1u << q) - 1u));                    // rank among the non-empty children
    const int c_q = q == 0 ? cnt[0] : q == 1 ? cnt[1] : q == 2 ? cnt[2] : cnt[3];
    int id = 0;
    if (mine) id = r < nf0 ? (int)L.free_list[nf0 - 1 - r] : min(alloc0 + (r - nf0), L.node_cap - 1);
    const int cid0 = __builtin_amdgcn_readlane(id, 0), cid1 = __builtin_amdgcn_readlane(id, 1),
              cid2 = __builtin_amdgcn_readlane(id, 2), cid3 = __builtin_amdgcn_readlane(id, 3);
    auto cid = [&](int qq) { return qq == 0 ? cid0 : qq == 1 ? cid1 : qq == 2 ? cid2 : cid3; };
    const int q_first = __ffs((int)ne_mask) - 1, q_last = 31 - __clz((int)ne_mask);
    const int first = cid(q_first), last = cid(q_last);
    const int after = head0 == node ? nv.next : head0;                  // what follows the children in the list
    if (mine) {
        const uint32_t lo = ne_mask & ((1u << q) - 1u), hi = ne_mask >> (q + 1);
        const int nxt = lo ? cid(31 - __clz((int)lo)) : after;          // towards the back: the previous non-empty quadrant
        const int prv = hi ? cid(q + 1 + (__ffs((int)hi) - 1)) : -1;    // towards the front: the next one
        QBox bx;
        bx.ulx = (short)((q & 1) ? midx : ulx); bx.uly = (short)((q & 2) ? midy : uly);
        bx.brx = (short)((q & 1) ? brx : midx); bx.bry = (short)((q & 2) ? bry : midy);
        const int cb = beg + (q > 0 ? cnt[0] : 0) + (q > 1 ? cnt[1] : 0) + (q > 2 ? cnt[2] : 0);
        L.seq[id] = (unsigned short)(seq0 + r);
        L.box[id] = bx; L.begin[id] = (unsigned short)cb; L.count[id] = (unsigned short)c_q;
        L.next[id] = (short)nxt; L.prev[id] = (short)prv;
        if (c_q > 1) {
            const int slot = n_exp + __popc(ex_mask & ((1u << q) - 1u));
            if (slot < L.node_cap) L.ea[slot] = exp_key(c_q, seq0 + r, id);
        }
    }
    // ---- the neighbours and the erased node (lane 0)
    if (lane == 0) {
        if (head0 != node) {
            L.prev[head0] = (short)first;
            L.next[nv.prev] = (short)nv.next;
            if (nv.next >= 0) L.prev[nv.next] = (short)nv.prev;
        } else if (nv.next >= 0) L.prev[nv.next] = (short)first;
        L.free_list[nf0 - take] = (short)node;
    }
    if (nv.next < 0) t.tail = head0 != node ? nv.prev : first;
    t.head = last;
    t.size += k - 1;
    t.seq = seq0 + k;
    t.n_free = nf0 - take + 1;
    t.n_alloc = min(alloc0 + fresh, L.node_cap);
    n_to_expand += g;
    n_exp = min(n_exp + g, max(n_exp, L.node_cap));                   // (the list stops growing at the capacity)
}

// ascending (size, creation order) -- CANONICAL (O1); keys are unique, so rank == final position
__device__ inline void exp_sort(const unsigned long long *in, unsigned long long *out, int n, int lane)
{
    for (int i = lane; i < n; i += 64) {
        const unsigned long long me = in[i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += in[j] < me;
        out[rank] = me;
    }
}

struct OrbDistArgs {
    OrbGeom g;
    const float4 *lvl_cand; const int *lvl_cnt; int cand_cap;
    uint2 *gkeys, *gtmp;                  // per-instance scratch, cand_cap entries each
    int *sel; int *sel_cnt; int sel_cap;  // outputs: selected candidate indices in list order
    int *overflow;
    int node_cap;
};

__global__ __launch_bounds__(64) void orb_distribute_kernel(OrbDistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_smem[];
    QLds L = qlds_carve(qt_smem, a.node_cap);
    const int lane = threadIdx.x;
    // workgroups are dispatched in index order: all level-0 instances (the longest: the largest quota) first,
    // the short top levels last, so the tail of the launch is made of short instances (longest-first packing;
    // with the level as the fast index the last round waited for level-0 trees: 2.4 ms instead of 1.3)
    const int b = blockIdx.x, l = blockIdx.y, inst = b * a.g.nlevels + l;
    const int nkeys = a.lvl_cnt[inst];
    const float4 *cand = a.lvl_cand + (int64_t)inst * a.cand_cap;
    int *sel = a.sel + (int64_t)inst * a.sel_cap;
    const int W = a.g.w[l], H = a.g.h[l];
    const int minX = 16, maxX = W - 16, minY = 16, maxY = H - 16, N = a.g.quota[l];
    const bool keys_global = nkeys > kLdsKeys;
    // global scratch of an instance: cand_cap x 8 bytes = one xy plane + one index plane
    KeyArr gk, gtmp;
    gk.xy = (uint32_t *)(a.gkeys + (int64_t)inst * a.cand_cap); gk.id = (uint16_t *)(gk.xy + a.cand_cap);
    gtmp.xy = (uint32_t *)(a.gtmp + (int64_t)inst * a.cand_cap); gtmp.id = (uint16_t *)(gtmp.xy + a.cand_cap);
    const KeyArr keys = keys_global ? gk : L.keys;
    QState t;
    t.n_free = 0; t.n_alloc = 0; t.head = t.tail = -1; t.size = 0; t.seq = 0; t.overflow = false;
    if (maxX <= minX || maxY <= minY || nkeys == 0) { if (lane == 0) a.sel_cnt[inst] = 0; return; }
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) { if (lane == 0) a.sel_cnt[inst] = 0; return; }
    const float hX = (float)(maxX - minX) / nIni;
    const unsigned long long lt = (1ull << lane) - 1ull;

    // root nodes: stable binning of the candidates by column strip
    int filled = 0;
    for (int i = 0; i < nIni && i < 64 && i < L.node_cap; i++) {
        const int start = filled;
        for (int base = 0; base < nkeys; base += 256) {
            float4 c4[4];                                     // four loads in flight
#pragma unroll
            for (int u = 0; u < 4; u++) c4[u] = cand[min(base + 64 * u + lane, nkeys - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = base + 64 * u + lane;
                const float4 c = c4[u];
                const bool mine = k < nkeys && min((int)(c.x / hX), nIni - 1) == i;
                const unsigned long long m = __ballot(mine);
                if (mine) keys.set(filled + __popcll(m & lt), make_uint2((uint32_t)(int)c.x | ((uint32_t)(int)c.y << 16), (uint32_t)k));
                filled += __popcll(m);
            }
        }
        const int id = qt_new(L, t, lane);
        if (lane == 0) {
            QBox bx; bx.ulx = (short)(int)(hX * (float)i); bx.uly = 0; bx.brx = (short)(int)(hX * (float)(i + 1));
            bx.bry = (short)(maxY - minY);
            L.box[id] = bx; L.begin[id] = (unsigned short)start; L.count[id] = (unsigned short)(filled - start);
        }
        qt_push_back(L, t, id, lane);
    }
    if (keys_global) __threadfence();
    for (int i = t.head; i >= 0;) {
        const int nx = rfl(L.next[i]);
        if (rfl(L.count[i]) == 0) qt_erase(L, t, i, lane);
        i = nx;
    }

    bool finish = false;
    int n_exp = 0;
    while (!finish) {
        int prevSize = t.size, nToExpand = 0, lit = t.head;
        n_exp = 0;
        while (lit >= 0) {
            const NodeView nv = qt_view(L, lit);
            if (nv.cnt == 1) { lit = nv.next; continue; }
            qt_split(L, t, keys, gtmp, keys_global, nv, lit, lane, n_exp, nToExpand);
            lit = nv.next;
            if (t.overflow) break;
        }
        if (t.overflow) break;
        if (t.size >= N || t.size == prevSize) finish = true;
        else if (t.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = t.size;
                const int n_prev = n_exp;
                n_exp = 0;
                exp_sort(L.ea, L.eb, n_prev, lane);
                for (int j = n_prev - 1; j >= 0; j--) {
                    const int nid = rfl((int)((uint32_t)L.eb[j] & 0xFFFu));
                    const NodeView nv = qt_view(L, nid);
                    qt_split(L, t, keys, gtmp, keys_global, nv, nid, lane, n_exp, nToExpand);
                    if (t.size >= N || t.overflow) break;
                }
                if (t.size >= N || t.size == prevSize || t.overflow) finish = true;
            }
        }
    }
    // leaves in list order -> best response of each (one leaf per lane)
    int *order = (int *)L.ea;
    int m = 0;
    for (int i = t.head; i >= 0 && m < a.sel_cap; i = rfl(L.next[i])) { if (lane == 0) order[m] = i; m++; }
    // responses (FAST scores, integers <= 255) as a byte table in the expansion list's second half, which is free now:
    // a leaf's keys are then compared from LDS (one dependent global load per key made this phase a tenth of the
    // kernel); candidate lists too long for the table keep reading the list in global memory
    uint8_t *resp8 = (uint8_t *)L.eb;
    const bool resp_lds = nkeys <= L.node_cap * 8;
    if (resp_lds)
        for (int i = lane; i < nkeys; i += 64) resp8[i] = (uint8_t)(int)cand[i].z;
    for (int j = lane; j < m; j += 64) {
        const int id = order[j], beg = L.begin[id], n = L.count[id];
        // keys of a node are in candidate order: the first maximum is the smallest index among the maxima
        int best = keys.id[beg];
        if (resp_lds) {
            int maxR = resp8[best];
            for (int k = 1; k < n; k++) {
                const int c = keys.id[beg + k], r = resp8[c];
                if (r > maxR) { best = c; maxR = r; }
            }
        } else {
            float maxR = cand[best].z;
            for (int k = 1; k < n; k++) {
                const int c = keys.id[beg + k];
                const float r = cand[c].z;
                if (r > maxR) { best = c; maxR = r; }
            }
        }
        sel[j] = best;
    }
    if (lane == 0) {
        a.sel_cnt[inst] = m;
        if (t.overflow) atomicOr(a.overflow + b, 1);
    }
}

// ---- DistributeOctTree, the passes in parallel over the nodes -----------------------------------
// ORB-SLAM2's loop (ORBextractor.cpp:553-700) is serial only in appearance.  A pass over the list divides EVERY node
// that holds more than one key; the children go to the FRONT of the list, so nothing a pass creates is visited by
// it: the divisions of a pass are independent, and the list it leaves behind is a pure function of the old list:
//     [ children of the LAST divided node (n4, n3, n2, n1 without the empty ones), ..., children of the FIRST ]
//     followed by the undivided nodes in their old order.
// The last phase ("sort the nodes to expand by size, divide the largest first, stop when the list holds N nodes")
// is the same with the sorted order as the processing order and a cut-off: a node's number of non-empty children
// is known from its keys per quadrant, the list size after the r-th division is a prefix sum, and the divisions up
// to the first one that reaches N are exactly those the serial loop performs.
// So a WORKGROUP OF FOUR WAVES owns an (image, level) -- a lone wave issues one instruction every ~5 cycles whatever
// it does, and the tree is ~10^5 instructions -- and a pass is
//   (1) the 4-way partition of every processed node's keys, which also counts them per quadrant, A LANE PER KEY: nodes
//       of <= 8 / 16 / 32 / 64 keys share a wave's lanes 8 / 4 / 2 / 1 at a time (the usual case from the third pass on:
//       a level-0 tree has ~250 nodes with a handful of keys each, and the serial kernel spent ~3 k cycles of dependent
//       instructions on each), ranks from four ballots masked to the node's lanes, in place; a wave together, in
//       registers and in place, on a larger node; the work items go round the four waves, the longest first;
//   (2) prefix sums over the processing order (workgroup scans through the LDS): children created before each node,
//       the cut-off (rounds of 256 nodes after the one that holds the cut-off are not even partitioned);
//   (3) the new list, written beside the old one (two node tables): children's creation numbers in creation order
//       (processing order, then quadrant), the nodes to expand next for the sort of the last phase.
// The list is an ARRAY in list order -- no links to chase.  A node past the cut-off has had its keys grouped by
// quadrant without being divided; that is harmless because the only thing read from an undivided node's keys is
// "largest response, first in candidate order among equals" == smallest candidate index among the maxima, which
// does not depend on their order.  Output and tie-breaks are those of orb_distribute_kernel (same keypoints byte
// for byte: the parity tests compare every image's), which stays for configurations whose node tables do not
// fit the LDS (and behind SVO_ORB_QT_SERIAL=1, to time one against the other).
#ifndef SVO_QP_WAVES
#define SVO_QP_WAVES 8
#endif
constexpr int kQpWaves = SVO_QP_WAVES;
constexpr int kQpThreads = 64 * kQpWaves;
constexpr int kQpScanBuf = 8;               // ints per scan buffer: one total per wave
static_assert(kQpWaves <= 8 && kQpThreads <= 1023, "scan buffers and the packed 10-bit class counters");
struct QpNode { short ulx, uly, brx, bry; unsigned short beg, cnt, seq, pad; };      // 16 bytes
struct QpLds {
    KeyArr keys;                            // kLdsKeys
    uint8_t *resp;                          // kLdsKeys: FAST score per candidate
    QpNode *cur, *nxt;                      // node tables (list order), node_cap each
    unsigned long long *xa;                 // nodes to expand: (count << 48 | creation number << 32 | list position)
    unsigned short *qc;                     // keys per quadrant of proc[r]: 4 per entry
    unsigned short *proc;                   // list positions in processing order
    unsigned short *kpre;                   // children created before proc[r] (exclusive prefix)
    unsigned short *divided;                // per list position: 1 if divided in this pass
    int *sbuf;                              // two scan buffers
    int *sh;                                // a few workgroup-wide scalars
};
__host__ __device__ inline size_t qplds_bytes(int node_cap)
{
    return (size_t)kLdsKeys * 7 + (size_t)node_cap * (16 * 2 + 8 + 8 + 2 * 3) + (size_t)kQpScanBuf * 2 * 4 + 32 + 16;
}
__device__ inline QpLds qplds_carve(uint8_t *smem, int node_cap)
{
    QpLds L;
    L.sbuf = (int *)smem; smem += (size_t)kQpScanBuf * 2 * 4;
    L.sh = (int *)smem; smem += 32;
    L.keys.xy = (uint32_t *)smem; smem += (size_t)kLdsKeys * 4;
    L.cur = (QpNode *)smem; smem += (size_t)node_cap * 16;
    L.nxt = (QpNode *)smem; smem += (size_t)node_cap * 16;
    L.xa = (unsigned long long *)smem; smem += (size_t)node_cap * 8;
    L.qc = (unsigned short *)smem; smem += (size_t)node_cap * 8;
    L.keys.id = (uint16_t *)smem; smem += (size_t)kLdsKeys * 2;
    L.proc = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.kpre = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.divided = (unsigned short *)smem; smem += (size_t)node_cap * 2;
    L.resp = smem;
    return L;
}

// exclusive prefix sum over the workgroup's threads in thread order; every thread gets the total.  Each wave scans its own
// lanes on the DPP network and only the wave totals go through the LDS: one barrier.  Consecutive calls must alternate
// between the two buffers (a thread may still be reading the previous call's totals when another starts the next).
__device__ __forceinline__ int blk_excl_scan(int v, int *buf, int lane, int wv, int &total)
{
    const int incl = wave_incl_scan(v);
    if (lane == 63) buf[wv] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kQpWaves; w++) {
        const int t = buf[w];
        all += t;
        if (w < wv) before += t;
    }
    total = rfl(all);
    return before + incl - v;
}

// a wave together on one node: stable 4-way partition in place (keys held in registers between the reads and the
// writes), nodes of more than 1024 keys through the instance's global scratch (at the node's own offset)
template <typename Fence>
__device__ inline void qp_partition(const KeyArr keys, const KeyArr gscratch, int beg, int n, int midx, int midy, int cnt[4], int lane, Fence fence)
{
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
    if (n <= 64) {
        const bool valid = lane < n;
        const uint2 k = valid ? keys.get(beg + lane) : make_uint2(0, 0);
        const int q = key_quadrant(k, midx, midy);
        const int r = chunk_rank(valid, q, lane, cnt);
        const int off = q == 0 ? 0 : (q == 1 ? cnt[0] : (q == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2]));
        if (valid) keys.set(beg + off + r, k);
    } else if (n <= 1024) {
        uint2 k[16];
        int q[16], r[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            k[c] = make_uint2(0, 0); q[c] = 0; r[c] = 0;
            if (64 * c >= n) continue;                        // (uniform: chunks past the node's last key cost nothing)
            k[c] = lane + 64 * c < n ? keys.get(beg + lane + 64 * c) : make_uint2(0, 0);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if (64 * c >= n) continue;
            q[c] = key_quadrant(k[c], midx, midy);
            r[c] = chunk_rank(lane + 64 * c < n, q[c], lane, cnt);
        }
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if (64 * c >= n) continue;
            const int off = q[c] == 0 ? 0 : (q[c] == 1 ? cnt[0] : (q[c] == 2 ? cnt[0] + cnt[1] : cnt[0] + cnt[1] + cnt[2]));
            if (lane + 64 * c < n) keys.set(beg + off + r[c], k[c]);
        }
    } else {
        for (int base = 0; base < n; base += 64) {
            const bool valid = base + lane < n;
            const uint2 k = valid ? keys.get(beg + base + lane) : make_uint2(0, 0);
            chunk_rank(valid, key_quadrant(k, midx, midy), lane, cnt);
        }
        int run[4] = {0, 0, 0, 0};
        const int o1 = cnt[0], o2 = cnt[0] + cnt[1], o3 = cnt[0] + cnt[1] + cnt[2];
        for (int base = 0; base < n; base += 64) {
            const bool valid = base + lane < n;
            const uint2 k = valid ? keys.get(beg + base + lane) : make_uint2(0, 0);
            const int q = key_quadrant(k, midx, midy);
            const int r = chunk_rank(valid, q, lane, run);
            const int off = q == 0 ? 0 : (q == 1 ? o1 : (q == 2 ? o2 : o3));
            if (valid) gscratch.set(beg + off + r, k);
        }
        __threadfence();
        for (int i = lane; i < n; i += 64) keys.set(beg + i, gscratch.get(beg + i));
    }
    fence();
}

// Diagnostic build (-DSVO_QT_STAMP, tools/gpu/qt_stamps.sh): the (image 0, level 0) instance prints the cycles its first
// wave spent per section.
#ifdef SVO_QT_STAMP
#define QT_AT(i) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime()); qt_acc[i] += now_ - qt_last; qt_last = now_; }
#else
#define QT_AT(i)
#endif

__global__ __launch_bounds__(kQpThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void orb_distribute_par_kernel(OrbDistArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qp_smem[];
    QpLds L = qplds_carve(qp_smem, a.node_cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x, l = blockIdx.y, inst = b * a.g.nlevels + l;
    const int nkeys = a.lvl_cnt[inst];
    const float4 *cand = a.lvl_cand + (int64_t)inst * a.cand_cap;
    int *sel = a.sel + (int64_t)inst * a.sel_cap;
    const int W = a.g.w[l], H = a.g.h[l];
    const int minX = 16, maxX = W - 16, minY = 16, maxY = H - 16, N = a.g.quota[l];
    const int cap = a.node_cap;
    if (maxX <= minX || maxY <= minY || nkeys == 0) { if (tid == 0) a.sel_cnt[inst] = 0; return; }
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni <= 0) { if (tid == 0) a.sel_cnt[inst] = 0; return; }
    const float hX = (float)(maxX - minX) / nIni;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // candidate lists longer than the LDS key arrays (cv::FAST is uncapped; the default capacity is 4 x max_keypoints per
    // level) are partitioned in the instance's global scratch instead: same code, the fences then cover global memory too
    const bool keys_global = nkeys > kLdsKeys;
    KeyArr gk, gt;
    gk.xy = (uint32_t *)(a.gkeys + (int64_t)inst * a.cand_cap); gk.id = (uint16_t *)(gk.xy + a.cand_cap);
    gt.xy = (uint32_t *)(a.gtmp + (int64_t)inst * a.cand_cap); gt.id = (uint16_t *)(gt.xy + a.cand_cap);
    const KeyArr keys = keys_global ? gk : L.keys;
    auto wfence = [&]() { if (keys_global) __threadfence(); wave_lds_fence(); };
    auto sync = [&]() { if (keys_global) __threadfence(); __syncthreads(); };
    int sbsel = 0;
    auto scan = [&](int v, int &total) -> int {
        int *bf = L.sbuf + kQpScanBuf * sbsel;
        sbsel ^= 1;
        return blk_excl_scan(v, bf, lane, wv, total);
    };
    bool overflow = false;
#ifdef SVO_QT_STAMP
    uint32_t qt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, qt_last = 0, qt_passes = 0;
    QT_AT(11) qt_acc[11] = 0;
#endif

    // ---- root nodes: stable binning of the candidates by column strip.  Each wave takes its part of the candidate list;
    // lane i keeps strip i's count in that part, then its fill position; two sweeps over the list (global memory,
    // eight loads in flight).  Empty roots are dropped, their creation numbers are not.
    const int nroot = min(min(nIni, 64), cap);
    int n = 0, seq = 0;
    {
        const int part = ((nkeys + kQpThreads - 1) / kQpThreads) * 64;
        const int kbeg = min(nkeys, wv * part), kend = min(nkeys, kbeg + part);
        int mine_cnt = 0, fill = 0;
        auto strip_of = [&](float x) { return min((int)(x / hX), nIni - 1); };
        auto count_chunk = [&](bool valid, int s) {
            for (int i = 0; i < nroot; i++) {
                const int c = __popcll(__ballot(valid && s == i));
                if (lane == i) mine_cnt += c;
            }
        };
        auto place_chunk = [&](bool valid, int s, int k, const float4 &c) {
            int dst = 0;
            for (int i = 0; i < nroot; i++) {
                const unsigned long long m = __ballot(valid && s == i);
                const int at = __builtin_amdgcn_readlane(fill, i);
                if (s == i) dst = at + __popcll(m & lt);
                if (lane == i) fill += __popcll(m);
            }
            if (valid && s < nroot) keys.set(dst, make_uint2((uint32_t)(int)c.x | ((uint32_t)(int)c.y << 16), (uint32_t)k));
            if (valid && !keys_global) L.resp[k] = (uint8_t)(int)c.z;
        };
        // a wave's part of a list that fits the LDS key arrays is read once and held in registers between the two sweeps
        constexpr int kRootChunks = (kLdsKeys + kQpThreads - 1) / kQpThreads;
        float4 held[kRootChunks];
        int held_s[kRootChunks];
        if (!keys_global) {
#pragma unroll
            for (int u = 0; u < kRootChunks; u++) held[u] = cand[min(kbeg + 64 * u + lane, nkeys - 1)];
#pragma unroll
            for (int u = 0; u < kRootChunks; u++) {
                held_s[u] = strip_of(held[u].x);
                if (kbeg + 64 * u < kend) count_chunk(kbeg + 64 * u + lane < kend, held_s[u]);
            }
        } else {
            for (int base = kbeg; base < kend; base += 512) {
                float cx[8];
#pragma unroll
                for (int u = 0; u < 8; u++) cx[u] = cand[min(base + 64 * u + lane, nkeys - 1)].x;
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (base + 64 * u < kend) count_chunk(base + 64 * u + lane < kend, strip_of(cx[u]));
            }
        }
        // [waves][64 strips], 2 KB at most: the second node table, the expansion list and the quadrant counts lie one after
        // the other (qplds_carve) and nothing has been written to them yet -- 32 bytes per node, node_cap >= 64
        int *rc = (int *)L.nxt;
        rc[wv * 64 + lane] = mine_cnt;
        __syncthreads();
        int tot = 0, mine_before = 0;
#pragma unroll
        for (int w = 0; w < kQpWaves; w++) {
            const int t = rc[w * 64 + lane];
            tot += t;
            if (w < wv) mine_before += t;
        }
        const int start = wave_incl_scan(tot) - tot;
        fill = start + mine_before;
        if (!keys_global) {
#pragma unroll
            for (int u = 0; u < kRootChunks; u++)
                if (kbeg + 64 * u < kend) place_chunk(kbeg + 64 * u + lane < kend, held_s[u], kbeg + 64 * u + lane, held[u]);
        } else {
            for (int base = kbeg; base < kend; base += 512) {
                float4 c4[8];
#pragma unroll
                for (int u = 0; u < 8; u++) c4[u] = cand[min(base + 64 * u + lane, nkeys - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (base + 64 * u < kend) place_chunk(base + 64 * u + lane < kend, strip_of(c4[u].x), base + 64 * u + lane, c4[u]);
            }
        }
        const bool have = lane < nroot && tot > 0;
        const unsigned long long hm = __ballot(have);
        if (wv == 0 && have) {
            QpNode nd;
            nd.ulx = (short)(int)(hX * (float)lane); nd.uly = 0; nd.brx = (short)(int)(hX * (float)(lane + 1)); nd.bry = (short)(maxY - minY);
            nd.beg = (unsigned short)start; nd.cnt = (unsigned short)tot; nd.seq = (unsigned short)lane; nd.pad = 0;
            L.cur[__popcll(hm & lt)] = nd;
        }
        n = __popcll(hm); seq = nroot;
    }
    sync();
    QT_AT(0)

    // One pass.  np nodes to process are listed in L.proc (list positions, processing order); limitN > 0: stop after the
    // division that brings the list to limitN nodes (the last phase), 0: divide them all.  Swaps the node tables, leaves
    // the children to expand next in L.xa (n_exp of them), updates n and seq.
    auto pass = [&](int np, int limitN, int &n_exp) {
        // (1) partition + keys per quadrant, (2) children before each processed node and the cut-off
        int carry = 0, ndiv = np, total_kids = 0;
        bool found = false;
        // [5 classes][min(threads, node_cap)] thread numbers of a round's nodes, in the table that is not the list (dead here; a
        // round has at most node_cap nodes, so the five lists of 16-bit entries fit its 16 bytes per node)
        unsigned short *lst = (unsigned short *)L.nxt;
        const int lstride = min(kQpThreads, cap);
        for (int r0 = 0; r0 < np && !found; r0 += kQpThreads) {
            const int r = r0 + tid;
            const bool act = r < np;
            const int mycnt = act ? (int)L.cur[L.proc[r]].cnt : 0;
            if (tid == 0) L.sh[0] = 0x7FFFFFFF;
            // the round's nodes by size class: <= 8, 16, 32, 64 keys share a wave's 64 lanes (8, 4, 2, 1 nodes at a time,
            // a lane per key); larger ones take a wave each.  Two scans of packed 10-bit counters give the class lists.
            const int cls = mycnt <= 8 ? 0 : mycnt <= 16 ? 1 : mycnt <= 32 ? 2 : mycnt <= 64 ? 3 : 4;
            int totA, totB;
            const int pa = scan(act && cls < 3 ? 1 << (10 * cls) : 0, totA);
            const int pb = scan(act && cls >= 3 ? 1 << (10 * (cls - 3)) : 0, totB);
            if (act) lst[cls * lstride + (cls < 3 ? (pa >> (10 * cls)) & 1023 : (pb >> (10 * (cls - 3))) & 1023)] = (unsigned short)tid;
            const int nc0 = totA & 1023, nc1 = (totA >> 10) & 1023, nc2 = totA >> 20, nc3 = totB & 1023, nhuge = totB >> 10;
            __syncthreads();
            QT_AT(1)
            // work items, the longest first: a large node each, then chunks of 1, 2, 4, 8 nodes; round-robin over the waves
            const int g3 = nhuge + nc3, g2 = g3 + ((nc2 + 1) >> 1), g1 = g2 + ((nc1 + 3) >> 2), g0 = g1 + ((nc0 + 7) >> 3);
            for (int g = wv; g < g0; g += kQpWaves) {
                if (g < nhuge) {
                    const int r2 = r0 + rfl((int)lst[4 * lstride + g]);
                    const QpNode nd = L.cur[rfl((int)L.proc[r2])];
                    const int ulx = rfl(nd.ulx), uly = rfl(nd.uly), brx = rfl(nd.brx), bry = rfl(nd.bry);
                    int c[4];
                    qp_partition(keys, gt, rfl(nd.beg), rfl(nd.cnt), ulx + ((brx - ulx + 1) >> 1), uly + ((bry - uly + 1) >> 1), c, lane, wfence);
                    if (lane == 0) *(uint2 *)(L.qc + 4 * r2) = make_uint2((uint32_t)c[0] | ((uint32_t)c[1] << 16), (uint32_t)c[2] | ((uint32_t)c[3] << 16));
                    continue;
                }
                const int c = g < g3 ? 3 : g < g2 ? 2 : g < g1 ? 1 : 0;
                const int ch = g - (c == 3 ? nhuge : c == 2 ? g3 : c == 1 ? g2 : g1), ncl = c == 3 ? nc3 : c == 2 ? nc2 : c == 1 ? nc1 : nc0;
                const int sh = 3 + c, S = 1 << sh;
                const int slot = lane >> sh, k = lane & (S - 1);
                const int idx = (ch << (3 - c)) + slot;
                const bool has = idx < ncl;
                int r2 = 0, beg = 0, cn = 0, midx = 0, midy = 0;
                if (has) {
                    r2 = r0 + lst[c * lstride + idx];
                    const QpNode nd = L.cur[L.proc[r2]];
                    beg = nd.beg; cn = nd.cnt;
                    midx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1); midy = nd.uly + ((nd.bry - nd.uly + 1) >> 1);   // ceil(d / 2)
                }
                const bool valid = has && k < cn;
                const uint2 key = valid ? keys.get(beg + k) : make_uint2(0, 0);
                const int q = key_quadrant(key, midx, midy);
                const unsigned long long m0 = __ballot(valid && q == 0), m1 = __ballot(valid && q == 1),
                                         m2 = __ballot(valid && q == 2), m3 = __ballot(valid && q == 3);
                const unsigned long long seg = c == 3 ? ~0ull : (((1ull << S) - 1ull) << (slot << sh));    // the lanes of this node
                const int n0 = __popcll(m0 & seg), n1 = __popcll(m1 & seg), n2 = __popcll(m2 & seg), n3 = __popcll(m3 & seg);
                const unsigned long long mq = q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;
                const int rank = __popcll(mq & seg & lt);                   // stable: keys of the same quadrant before this one
                const int off = q == 0 ? 0 : q == 1 ? n0 : q == 2 ? n0 + n1 : n0 + n1 + n2;
                if (valid) keys.set(beg + off + rank, key);                 // (in place: the wave has read every key of the chunk by now)
                if (has && k == 0) *(uint2 *)(L.qc + 4 * r2) = make_uint2((uint32_t)n0 | ((uint32_t)n1 << 16), (uint32_t)n2 | ((uint32_t)n3 << 16));
            }
            sync();
            QT_AT(2)
            int kids = 0;
            if (act) {
                const uint2 qv = *(const uint2 *)(L.qc + 4 * r);
                kids = ((qv.x & 0xFFFFu) != 0) + ((qv.x >> 16) != 0) + ((qv.y & 0xFFFFu) != 0) + ((qv.y >> 16) != 0);
            }
            int ktot;
            const int pre = carry + scan(kids, ktot);
            if (act) L.kpre[r] = (unsigned short)pre;
            carry += ktot;
            total_kids = carry;
            if (limitN > 0) {
                // list size after dividing nodes 0..r: n + (children of 0..r) - (r + 1); the first r that reaches limitN
                if (act && n + pre + kids - (r + 1) >= limitN) atomicMin(L.sh, (r << 16) | (pre + kids));
                __syncthreads();
                const int first = rfl(L.sh[0]);
                if (first != 0x7FFFFFFF) { ndiv = (first >> 16) + 1; total_kids = first & 0xFFFF; found = true; }
            }
        }
        sync();
        const int n_new = n + total_kids - ndiv;
        if (n_new > cap || seq + total_kids > 65535) { overflow = true; return; }
        for (int i = tid; i < n; i += kQpThreads) L.divided[i] = 0;
        __syncthreads();
        for (int r = tid; r < ndiv; r += kQpThreads) L.divided[L.proc[r]] = 1;
        QT_AT(3)
        // (3) the children of the divided nodes
        int xcarry = 0;
        for (int r0 = 0; r0 < ndiv; r0 += kQpThreads) {
            const int r = r0 + tid;
            const bool act = r < ndiv;
            int nx = 0, blk = 0, kids = 0, pre = 0;
            int cq[4] = {0, 0, 0, 0};
            if (act) {
                const QpNode nd = L.cur[L.proc[r]];
                const uint2 qv = *(const uint2 *)(L.qc + 4 * r);
                cq[0] = (int)(qv.x & 0xFFFFu); cq[1] = (int)(qv.x >> 16); cq[2] = (int)(qv.y & 0xFFFFu); cq[3] = (int)(qv.y >> 16);
                pre = L.kpre[r];
                kids = (cq[0] > 0) + (cq[1] > 0) + (cq[2] > 0) + (cq[3] > 0);
                blk = total_kids - (pre + kids);        // the block of node r starts where the children of the nodes processed AFTER it end
                const int midx = nd.ulx + ((nd.brx - nd.ulx + 1) >> 1), midy = nd.uly + ((nd.bry - nd.uly + 1) >> 1);
                int before = 0, begk = nd.beg;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (cq[q] > 0) {
                        QpNode ch;
                        ch.ulx = (short)((q & 1) ? midx : nd.ulx); ch.uly = (short)((q & 2) ? midy : nd.uly);
                        ch.brx = (short)((q & 1) ? nd.brx : midx); ch.bry = (short)((q & 2) ? nd.bry : midy);
                        ch.beg = (unsigned short)begk; ch.cnt = (unsigned short)cq[q]; ch.seq = (unsigned short)(seq + pre + before); ch.pad = 0;
                        L.nxt[blk + (kids - 1 - before)] = ch;              // n4 first ... n1 last
                        nx += cq[q] > 1;
                        before++;
                    }
                    begk += cq[q];
                }
            }
            // nodes to expand next, in creation order (the sort of the last phase is by (count, creation number))
            int xtot;
            const int xpos = xcarry + scan(nx, xtot);
            xcarry += xtot;
            if (act && nx) {
                int before = 0, w = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (cq[q] > 0) {
                        if (cq[q] > 1) {
                            L.xa[xpos + w] = ((unsigned long long)(((uint32_t)cq[q] << 16) | (uint32_t)(seq + pre + before)) << 32) |
                                             (unsigned long long)(blk + (kids - 1 - before));
                            w++;
                        }
                        before++;
                    }
                }
            }
        }
        __syncthreads();
        QT_AT(4)
        // the undivided nodes keep their order behind the children
        int ucarry = 0;
        for (int i0 = 0; i0 < n; i0 += kQpThreads) {
            const int i = i0 + tid;
            const bool und = i < n && L.divided[i] == 0;
            int utot;
            const int rank = ucarry + scan(und ? 1 : 0, utot);
            ucarry += utot;
            if (und) L.nxt[total_kids + rank] = L.cur[i];
        }
        sync();
        QpNode *sw = L.cur; L.cur = L.nxt; L.nxt = sw;
        n = n_new; seq += total_kids; n_exp = xcarry;
        QT_AT(5)
#ifdef SVO_QT_STAMP
        qt_passes++;
#endif
    };

    bool finish = false;
    int n_exp = 0;
    while (!finish && !overflow) {
        const int prevSize = n;
        // the bulk pass: every node with more than one key, in list order
        int np = 0;
        for (int i0 = 0; i0 < n; i0 += kQpThreads) {
            const int i = i0 + tid;
            const bool ex = i < n && L.cur[i].cnt > 1;
            int ptot;
            const int r = np + scan(ex ? 1 : 0, ptot);
            np += ptot;
            if (ex) L.proc[r] = (unsigned short)i;
        }
        sync();
        QT_AT(6)
        int nToExpand = 0;
        if (np > 0) pass(np, 0, nToExpand);
        if (overflow) break;
        if (n >= N || n == prevSize) finish = true;
        else if (n + nToExpand * 3 > N) {
            n_exp = nToExpand;
            while (!finish && !overflow) {
                const int prev2 = n, ne = n_exp;
                if (ne == 0) break;
                // processing order: descending (count, creation number).  Rank sort in registers: an entry per thread
                // against every entry, broadcast lane by lane (v_readlane) from one register per 64
                for (int i0 = 0; i0 < ne; i0 += kQpThreads) {
                    const unsigned long long e = L.xa[min(i0 + tid, ne - 1)];
                    const uint32_t me = (uint32_t)(e >> 32);
                    int rank = 0;
                    for (int j0 = 0; j0 < ne; j0 += 64) {
                        // (lanes past the end hold 0, which is larger than nothing: all 64 are compared, unrolled)
                        const uint32_t x = j0 + lane < ne ? (uint32_t)(L.xa[j0 + lane] >> 32) : 0u;
#pragma unroll
                        for (int t = 0; t < 64; t++) rank += (uint32_t)__builtin_amdgcn_readlane((int)x, t) > me;
                    }
                    if (i0 + tid < ne) L.proc[rank] = (unsigned short)(uint32_t)e;
                }
                sync();
                QT_AT(7)
                pass(ne, N, n_exp);
                if (n >= N || n == prev2) finish = true;
            }
            finish = true;
        }
    }
    // ---- leaves in list order -> best response of each (one leaf per thread): the largest, the smallest candidate index
    // among equals (== the first in candidate order, ORBextractor.cpp:703-722)
    const int m = min(n, a.sel_cap);
    for (int j = tid; j < m; j += kQpThreads) {
        const QpNode nd = L.cur[j];
        int best = 0x7FFFFFFF, maxR = -1;
        for (int k0 = 0; k0 < nd.cnt; k0 += 4) {
            int id[4], rs[4];
#pragma unroll
            for (int u = 0; u < 4; u++) id[u] = keys.id[nd.beg + min(k0 + u, nd.cnt - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) rs[u] = keys_global ? (int)cand[id[u]].z : (int)L.resp[id[u]];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (rs[u] > maxR || (rs[u] == maxR && id[u] < best)) { best = id[u]; maxR = rs[u]; }
        }
        sel[j] = best;
    }
    QT_AT(8)
#ifdef SVO_QT_STAMP
    if (tid == 0 && inst == 0)
        printf("qt nkeys %d n %d passes %u | roots %u small %u big %u scan %u kids %u undiv %u proc %u sort %u final %u\n", nkeys, n, qt_passes,
               qt_acc[0], qt_acc[1], qt_acc[2], qt_acc[3], qt_acc[4], qt_acc[5], qt_acc[6], qt_acc[7], qt_acc[8]);
#endif
    if (tid == 0) {
        a.sel_cnt[inst] = m;
        if (overflow) atomicOr(a.overflow + b, 1);
    }
}

// ---- blur (7x7, sigma 2, reflect-101; integer kernel, (sum + 2^15) >> 16, saturated) -------------
// One pass: a thread owns 4 adjacent columns of a 28-row band.  A row's four 7-tap sums are two
// v_dot4_u32_u8 each on byte windows cut from three aligned dwords (reflect-101 at the image edges by
// index arithmetic in the few threads that touch them); the seven most recent row sums slide through registers for the
// column pass (28-row band), so neither the int intermediate image nor a second launch exists.
constexpr int kBlurRows = kBlurRowsPerThread;     // 4 x 7: the row ring rotates with static indices
__global__ __launch_bounds__(256) void orb_blur_kernel(OrbGeom g, const uint8_t *__restrict__ slots, int64_t slot_stride,
                                                       uint8_t *__restrict__ blur, int64_t blur_img_stride, OrbL0 z)
{
    const int b = blockIdx.z;
    const int l = level_of_block(g.blur_blk, g.nlevels, blockIdx.y);      // blockIdx.y = row band over all levels
    const bool in_place = l == 0 && z.img != nullptr;                     // level 0 = the input image itself
    const int w = g.w[l], h = g.h[l], pitch = in_place ? z.pitch : g.pitch[l];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y0 = ((blockIdx.y - g.blur_blk[l]) * 4 + threadIdx.y) * kBlurRows;
    if (x0 >= w || y0 >= h) return;
    const uint8_t *lvl = in_place ? orb_level0(z, b) : slots + (int64_t)b * slot_stride + g.origin[l];
    uint8_t *dst = blur + (int64_t)b * blur_img_stride + g.blur_off[l] + x0;
    const int bp = g.bpitch[l];
    // BORDER_REFLECT_101 is applied HERE (ORBextractor.cpp:1034-1035 blurs the un-bordered clone of the level):
    // rows by reflecting the row index, columns only in the threads whose 12-byte window leaves the row -- the
    // first thread of a row and the last one or two.  No kernel writes a frame around the ORB levels any more
    // (the border launch was 0.3-0.5 ms per 256 pairs for bytes only this kernel ever read).
    const bool edge = x0 == 0 || x0 + 8 > w;
    const uint32_t k0123 = (uint32_t)g.gk[0] | ((uint32_t)g.gk[1] << 8) | ((uint32_t)g.gk[2] << 16) | ((uint32_t)g.gk[3] << 24);
    const uint32_t k456 = (uint32_t)g.gk[4] | ((uint32_t)g.gk[5] << 8) | ((uint32_t)g.gk[6] << 16);
    // byte shuffles that put the reflected columns into an edge thread's 12-byte window (columns x0-4 .. x0+7 =
    // window bytes 0..11): every column a valid output needs (<= w+2, >= -3) reflects to a column inside the same
    // window.  Window byte i takes byte sidx(i); two v_perm per dword (one over d0|d1, one bringing in d2).
    uint32_t selA[3], selB[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t sa = 0, sb = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = 4 * k + q, col = x0 - 4 + i;
            int s = i;
            if (col < 0 || col >= w) {
                const int r = col < 0 ? -col : 2 * w - 2 - col;
                const int j = r - (x0 - 4);
                if (r >= 0 && r < w && j >= 0 && j < 12) s = j;
            }
            // stage 1: t = perm(d1, d0, selA): bytes of d0 (0-3) / d1 (4-7); a d2 source leaves a placeholder
            sa |= (uint32_t)(s < 8 ? s : 0) << (8 * q);
            // stage 2: out = perm(d2, t, selB): t's byte q (index q) or d2's byte (4 + s - 8)
            sb |= (uint32_t)(s < 8 ? q : 4 + (s - 8)) << (8 * q);
        }
        selA[k] = sa; selB[k] = sb;
    }
    const bool wave_edge = __builtin_amdgcn_ballot_w64(edge) != 0;       // (uniform: decides the code path of the whole wave)
    auto rowsum = [&](const uint32_t (&d)[3], uint32_t (&out)[4]) {      // d: columns x0-4 .. x0+7 of one row
#pragma unroll
        for (int o = 0; o < 4; o++) {                               // taps of column x0+o: bytes o+1 .. o+7
            const uint32_t lo = o == 3 ? d[1] : __builtin_amdgcn_alignbyte(d[1], d[0], o + 1);
            const uint32_t hi = o == 3 ? d[2] : __builtin_amdgcn_alignbyte(d[2], d[1], o + 1);
            out[o] = __builtin_amdgcn_udot4(lo, k0123, __builtin_amdgcn_udot4(hi, k456, 0u, false), false);
        }
    };
    auto load_row = [&](int yy, uint32_t (&d)[3]) {
        yy = yy < 0 ? -yy : (yy >= h ? 2 * h - 2 - yy : yy);        // reflect-101 (one reflection: |offset| < 32 <= h)
        yy = min(max(yy, 0), h - 1);
        // (a thread at a row end reads a few bytes past it -- the slot's unused frame area, or an input row's padding --; the
        //  dword LEFT of a row's first one is never needed, the reflection replaces every byte of it: the row's first thread
        //  reads its own dword twice instead, so that row 0 of an input image read in place stays inside the buffer)
        const uint32_t *p = (const uint32_t *)(lvl + (int64_t)yy * pitch + x0);
        d[0] = p[x0 == 0 ? 0 : -1]; d[1] = p[0]; d[2] = p[1];
        if (wave_edge) {
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2];
#pragma unroll
            for (int k = 0; k < 3; k++) d[k] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, selA[k]), selB[k]);
        }
    };
    // ring of the seven most recent row sums: row y0-3+i lives in win[i % 7] (static indices: the
    // band is walked in fully unrolled groups of seven rows).  The seven source rows of a group are
    // requested together, before the first of them is used (one row per step made every thread a chain of
    // 34 dependent memory round trips: 0.85 ms per 256 pairs, four times either of the kernel's roofs);
    // (rows past the band's last one reflect back into the image: loaded, never used for an output row)
    uint32_t win[7][4];
    {
        uint32_t d[6][3];
#pragma unroll
        for (int r = 0; r < 6; r++) load_row(y0 - 3 + r, d[r]);
#pragma unroll
        for (int r = 0; r < 6; r++) rowsum(d[r], win[r]);
    }
    const int rows = min(kBlurRows, h - y0);
    for (int k = 0; k < kBlurRows / 7; k++) {
        if (7 * k >= rows) return;
        uint32_t d[7][3];
#pragma unroll
        for (int j = 0; j < 7; j++) load_row(y0 + 7 * k + j + 3, d[j]);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int yy = 7 * k + j;
            rowsum(d[j], win[(j + 6) % 7]);                     // row index yy + 6 in the ring
            uint32_t v4 = 0;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                uint32_t v = 1u << 15;
#pragma unroll
                for (int r = 0; r < 7; r++) v += win[(j + r) % 7][o] * (uint32_t)g.gk[r];
                v >>= 16;
                v4 |= (v > 255u ? 255u : v) << (8 * o);
            }
            if (yy < rows) *(uint32_t *)(dst + (int64_t)(y0 + yy) * bp) = v4;     // rows are padded to whole dwords: one aligned store
        }
    }
}

// ---- orientation + descriptor + final keypoint record -------------------------------------------
__device__ inline float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
                p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.2204460492503131e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct OrbDescArgs {
    OrbGeom g;
    const uint8_t *slots; int64_t slot_stride;
    const uint8_t *blur; int64_t blur_img_stride;
    OrbL0 z;                                                         // level 0 read in place (z.img != null)
    const float4 *lvl_cand; int cand_cap;
    const int *sel; const int *sel_cnt; int sel_cap;
    svo_keypoint *kps; uint8_t *desc; int *n_out; int out_cap;       // per image: out_cap keypoints
    int *overflow;                                                   // per image
    int blocks_per_img, n_img;                                       // one-dimensional grid of blocks_per_img x n_img blocks
};

// TWO keypoints per wave (one per 32-lane half): the orientation sums use 31 lanes and the
// descriptor 32, so with one keypoint per wave half of every instruction in those phases -- and
// all of the per-keypoint scalar work (level search, fastAtan2, the f64 sincos, the record) -- was
// issued for idle lanes; the kernel is VALU bound (80 % of the issue roof).  The 749 patch pixels
// are reduced with exact integer sums.
__device__ __forceinline__ int half_sum_i32(int v)               // sum over the 32 lanes of this half-wave
{
    v = row_allsum_i32(v);
    return v + __shfl_xor(v, 16, 64);
}
__global__ __launch_bounds__(256) void orb_describe_kernel(OrbDescArgs a)
{
    int b, bx;                                                           // image by XCD (xcd_image_block), block of the image
    xcd_image_block(blockIdx.x, a.blocks_per_img, a.n_img, b, bx);
    const int lane = threadIdx.x & 63, half = lane >> 5, sl = lane & 31;
    const int gidx = (bx * 4 + (threadIdx.x >> 6)) * 2 + half;           // keypoint index in the image's output
    // locate level: prefix over the per-level selected counts
    // (the per-level counts are fetched by eight lanes at once and handed round as scalars: read one after the
    //  other -- once for the total, again for the level search -- they were a chain of ~15 dependent global loads
    //  at the head of every wave)
    int l = 0, base = 0, total = 0;
    int cl[kOrbMaxLevels];
    {
        const int mine = (lane & 7) < a.g.nlevels ? a.sel_cnt[b * a.g.nlevels + (lane & 7)] : 0;
        static_assert(kOrbMaxLevels <= 8, "one lane per level");
#pragma unroll
        for (int q = 0; q < kOrbMaxLevels; q++) { cl[q] = __builtin_amdgcn_readlane(mine, q); total += q < a.g.nlevels ? cl[q] : 0; }
    }
    if (bx == 0 && threadIdx.x == 0) {
        a.n_out[b] = min(total, a.out_cap);
        if (total > a.out_cap) atomicOr(a.overflow + b, 4);          // more keypoints than max_keypoints
    }
    __shared__ int4 s_lv[kOrbMaxLevels];                                 // origin, pitch, blurred pitch, blurred offset of a level
    if (threadIdx.x < kOrbMaxLevels)
        s_lv[threadIdx.x] = threadIdx.x == 0 && a.z.img ? make_int4(0, a.z.pitch, a.g.bpitch[0], (int)a.g.blur_off[0])      // level 0 in place: see img_base
                                                       : make_int4((int)a.g.origin[threadIdx.x], a.g.pitch[threadIdx.x], a.g.bpitch[threadIdx.x], (int)a.g.blur_off[threadIdx.x]);
    __syncthreads();
    const bool valid = gidx < total && gidx < a.out_cap;
    if (__ballot(valid) == 0ull) return;
    const int gq = valid ? gidx : 0;                                  // an idle half shadows keypoint 0, stores nothing
    {
        // (branch-free search over the register copies: per-lane level, no memory in the loop)
        int run = 0;
#pragma unroll
        for (int q = 0; q < kOrbMaxLevels - 1; q++) {
            const bool beyond = q < a.g.nlevels - 1 && gq >= run + cl[q];        // the keypoint lies past level q
            if (beyond) { l = q + 1; base = run + cl[q]; }
            run += cl[q];
        }
    }
    const int inst = b * a.g.nlevels + l;
    const float4 cand = a.lvl_cand[(int64_t)inst * a.cand_cap + a.sel[(int64_t)inst * a.sel_cap + (gq - base)]];
    const float x = cand.x + 16.f, y = cand.y + 16.f;                // += minBorderX / minBorderY
    const int ix = __float2int_rn(x), iy = __float2int_rn(y);
    // the level's geometry from a small LDS table (the level differs from lane half to lane half: indexing the
    // kernel-argument arrays with it is a chain of selects per value, ~50 instructions)
    const int4 lv = s_lv[l];
    const int pitch = lv.y, bp = lv.z;
    // Both gathers (749-pixel circular patch of the level, <= 39x39 footprint of the rotated pattern
    // in the blurred level) go through LDS: a half-wave fetches its two windows as aligned dwords with
    // all loads in flight at once, then reads single bytes from LDS.
    __shared__ uint32_t s_raw[8][31 * 9], s_blr[8][39 * 11];
    const int hw = (threadIdx.x >> 6) * 2 + half;
    uint32_t *raw = s_raw[hw], *blr = s_blr[hw];
    // (addresses = a base that is uniform over the wave + a 32-bit offset per lane: an image's slot is a few MB.  The keypoints of
    // a level 0 that is read in place take their windows from the input image instead: keypoints are ordered by level, so a wave
    // is all level 0 or has none of it -- the base stays in scalar registers -- except the one wave per image that straddles the
    // boundary, which runs the fetch twice, each half-wave in its pass)
    const uint8_t *slot_base = a.slots + (int64_t)b * a.slot_stride;               // 4-byte aligned rows
    const uint8_t *lvl0 = a.z.img ? orb_level0(a.z, b) : nullptr;
    const unsigned long long m0 = lvl0 ? __ballot(l == 0) : 0ull;
    const bool mixed = m0 != 0ull && m0 != ~0ull;
    const int rx0 = (ix - 15) & ~3, roff = (ix - 15) - rx0;
    // a lane keeps ONE dword column of the window and walks down the rows (27 of the 32 lanes: 9 columns x 3 rows per
    // step): the address and the LDS index advance by a constant, no division or multiplication per load
    // (straight-line code, so that these loads and the blurred window's below are all in flight together; the straddling wave's
    // other half is fetched by the rare branch behind the LDS stores)
    const bool from0 = m0 != 0ull;                                    // a mixed wave serves its level-0 half first
    const uint8_t *img_base = from0 ? lvl0 : slot_base;
    const bool mine = sl < 27 && (!mixed || l == 0);
    const int rc = sl % 9, rrw = sl / 9;
    uint32_t vraw[11];                                                // (all the loads in flight before the first LDS store)
    {
        uint32_t off = (uint32_t)lv.x + (uint32_t)((iy - 15 + rrw) * pitch + rx0 + 4 * rc);
#pragma unroll
        for (int t = 0; t < 11; t++) {
            vraw[t] = mine && rrw + 3 * t < 31 ? *(const uint32_t *)(img_base + off) : 0u;
            off += 3u * (uint32_t)pitch;
        }
    }
    const uint8_t *blur_base = a.blur + (int64_t)b * a.blur_img_stride;          // 4-byte aligned rows, bp apart
    const int bx0 = (ix - 19) & ~3, boff = (ix - 19) - bx0;          // (the 39 x 39 footprint lies inside the level: >= 19 pixels from its edges)
    {
        const int c = sl % 11, rr = sl / 11;                          // 22 lanes: 11 columns x 2 rows per step
        uint32_t off = (uint32_t)lv.w + (uint32_t)((iy - 19 + rr) * bp + bx0 + 4 * c);
        int li = rr * 11 + c;
        uint32_t v[20];
#pragma unroll
        for (int t = 0; t < 20; t++) {
            v[t] = sl < 22 && rr + 2 * t < 39 ? *(const uint32_t *)(blur_base + off) : 0u;
            off += 2u * (uint32_t)bp;
        }
#pragma unroll
        for (int t = 0; t < 20; t++) {
            if (sl < 22 && rr + 2 * t < 39) blr[li] = v[t];
            li += 22;
        }
    }
    {
        int li = rrw * 9 + rc;
#pragma unroll
        for (int t = 0; t < 11; t++) {
            if (mine && rrw + 3 * t < 31) raw[li] = vraw[t];
            li += 27;
        }
    }
    if (__builtin_expect(mixed, 0)) {                                 // the half-wave whose keypoint is NOT at level 0: from the slot
        const bool mine2 = sl < 27 && l != 0;
        uint32_t off = (uint32_t)lv.x + (uint32_t)((iy - 15 + rrw) * pitch + rx0 + 4 * rc);
        int li = rrw * 9 + rc;
#pragma nounroll
        for (int t = 0; t < 11; t++) {
            if (mine2 && rrw + 3 * t < 31) raw[li] = *(const uint32_t *)(slot_base + off);
            off += 3u * (uint32_t)pitch;
            li += 27;
        }
    }
    wave_lds_fence();
    // IC_Angle: m10 = sum u * I, m01 = sum v * I over the circular patch (rows v = -15..15).  A lane takes a row: its 31
    // bytes as eight dwords shifted to start at u = -15, the pixels outside |u| <= umax[|v|] masked off, then four
    // bytes per instruction: v_sad_u8 against 0 adds them up, v_dot4_u32_u8 against (j, j+1, j+2, j+3) gives
    // sum (u + 15) * I (exact integers: sum u * I = that - 15 sum I)
    int m10 = 0, m01 = 0;
    if (sl < 31) {
        const int v = sl - 15, d = a.g.umax[v < 0 ? -v : v];
        const uint32_t *rw = raw + sl * 9;
        uint32_t dw[9];
#pragma unroll
        for (int k = 0; k < 9; k++) dw[k] = rw[k];
        const uint32_t bits = (2u << (15 + d)) - (1u << (15 - d));           // bit j set <=> column u = j - 15 is inside the patch
        uint32_t s = 0, sw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t e = __builtin_amdgcn_alignbit(dw[k + 1], dw[k], 8 * roff);          // bytes j = 4k .. 4k + 3
            const uint32_t px = e & ((((bits >> (4 * k)) & 15u) * 0x00204081u & 0x01010101u) * 0xFFu);
            const uint32_t wj = (uint32_t)(4 * k) * 0x01010101u + 0x03020100u;
            s = __builtin_amdgcn_sad_u8(px, 0u, s);
            sw = __builtin_amdgcn_udot4(px, wj, sw, false);
        }
        m10 = (int)sw - 15 * (int)s; m01 = v * (int)s;
    }
    m10 = half_sum_i32(m10); m01 = half_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // computeOrbDescriptor on the blurred level
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = angle * factorPI;
    double sd, cd;                                 // one shared argument reduction for both
    sincos((double)ang, &sd, &cd);
    const float ca = (float)cd, sb = (float)sd;
    {
        const signed char *pat = c_pattern + sl * 32;
        auto sample = [&](int dyy, int dxx) -> int { return ((const uint8_t *)(blr + (dyy + 19) * 11))[boff + dxx + 19]; };
        int val = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float xa = (float)pat[4 * k], ya = (float)pat[4 * k + 1], xb = (float)pat[4 * k + 2], yb = (float)pat[4 * k + 3];
            const int t0 = sample(__float2int_rn(xa * sb + ya * ca), __float2int_rn(xa * ca - ya * sb));
            const int t1 = sample(__float2int_rn(xb * sb + yb * ca), __float2int_rn(xb * ca - yb * sb));
            val |= (t0 < t1) << k;
        }
        if (valid) a.desc[((int64_t)b * a.out_cap + gidx) * 32 + sl] = (uint8_t)val;
    }
    if (sl == 0 && valid) {
        svo_keypoint kp;
        const float sc = a.g.scale[l];
        kp.x = l != 0 ? x * sc : x; kp.y = l != 0 ? y * sc : y;
        kp.size = (float)(int)(31 * sc); kp.angle = angle; kp.response = cand.z; kp.octave = l; kp.class_id = -1;
        a.kps[(int64_t)b * a.out_cap + gidx] = kp;
    }
}

// ---- BruteForce-Hamming match on the matrix cores: first minimum over all train rows ----------
// All-pairs Hamming distance IS a dense contraction: with the query bits as +-1 and the train bits as 0 / 1,
//   dot(q, t) = 2 pop(q & t) - pop(t)   and   hamming(q, t) = pop(q) + pop(t) - 2 pop(q & t) = pop(q) - dot(q, t),
// so for a query the nearest train row is the one of LARGEST dot product, and the distance follows from it --
// exact integers, the same winner and the same distance as the popcount scan this replaces (which sat at its
// VALU roof: 8 xor + 8 v_bcnt per descriptor pair, 1.2 ms per 256 pairs).  v_mfma_i32_32x32x32_i8 does 32 x 32
// descriptor pairs x 32 bits per instruction (operand lane maps checked with exact integers:
// tools/gpu/mfma_i8_probe.hip):
//   * workgroup = 8 waves = 256 queries; a wave keeps its 32 queries as the B operand of all eight k-steps in
//     32 registers (bits spread to bytes with one multiply per nibble, once per wave);
//   * train rows are read PACKED (32 bytes each: the traffic of the old kernel / 8) 64 at a time, spread to 0 / 1
//     bytes by the staging threads (a dword of bits -> 32 bytes) into a double-buffered LDS tile whose 16-byte
//     chunks are XOR-swizzled by the row (conflict-free ds_write_b128 / ds_read_b128), one barrier per tile;
//   * A = 32 train rows, so a lane holds 16 train rows x ONE query column: the running maximum of
//     ((dot + 256) << 16 | 0xFFFF - train index) stays inside the lane (16 v_lshl_or + 8 v_max3 per 32 x 32 block,
//     no cross-lane step until the two lane halves are merged at the very end); the key's low half makes the
//     maximum the FIRST minimum of the distance, as cv::BFMatcher returns it.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
constexpr int kMmT = 64;                         // train rows per LDS tile
// NW waves = 32 NW queries per workgroup: 8 for batches (a train tile is spread once per 256 queries), 4 for launches
// that leave the chip mostly empty (the online pair: one wave per SIMD runs its tile loop twice as fast as two)

// 4 bits -> 4 bytes of 0 / 1 (bit b of the nibble -> byte b)
__device__ __forceinline__ uint32_t spread_nibble(uint32_t n) { return (n * 0x00204081u) & 0x01010101u; }

template <int NW>
__global__ __launch_bounds__(64 * NW) void orb_match_kernel(const uint8_t *q, const int *nq_p, int nq_fixed, int64_t q_stride,
                                                        const uint8_t *t, const int *nt_p, int nt_fixed, int64_t t_stride,
                                                        int n_stride, int *idx, float *dist, int64_t out_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][kMmT * 256];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per-item row counts sit n_stride ints apart (image slots: left / right of consecutive frames)
    const int nq = nq_p ? nq_p[(int64_t)b * n_stride] : nq_fixed, nt = nt_p ? nt_p[(int64_t)b * n_stride] : nt_fixed;
    constexpr int kMmQ = 32 * NW, kStage = 8 / NW;                   // staging items (train row, packed dword) per thread
    static_assert(NW == 8 || NW == 4, "512 staging items per tile over 64 NW threads");
    const int i0 = blockIdx.x * kMmQ;
    if (i0 >= nq) return;                                            // uniform over the workgroup
    const int r = lane & 31, h = lane >> 5;
    const int iq = i0 + wave * 32 + r;                               // this lane's query (both lane halves: the same one)
    // ---- B operand: the query's bits [32 s + 16 h, + 16) of every k-step s as +-1 bytes; pop(q)
    v4i_t Bq[8];
    int popq = 0;
    {
        const uint32_t *qr = (const uint32_t *)(q + (int64_t)b * q_stride + (int64_t)min(iq, nq - 1) * 32);   // duplicates of the last row are not stored
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const uint32_t w = qr[s];
            popq += __popc(w);
            const uint32_t half = (w >> (16 * h)) & 0xFFFFu;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t x = spread_nibble((half >> (4 * c)) & 15u);          // 0 / 1 bytes
                Bq[s][c] = (int)(x | ((x ^ 0x01010101u) * 0xFFu));                  // 1 -> 0x01, 0 -> 0xFF (= -1)
            }
        }
    }
    // per-lane constants of the result rows: register g of a 32 x 32 block holds train row (g & 3) + 8 (g >> 2) + 4 h
    uint32_t Crow[16];
#pragma unroll
    for (int g = 0; g < 16; g++) Crow[g] = 0xFFFFu - (uint32_t)((g & 3) + 8 * (g >> 2) + 4 * h);
    const uint8_t *tb = t + (int64_t)b * t_stride;
    // ---- staging: item = (train row of the tile, packed dword w) -> 32 bytes of 0 / 1 = k-step w; kStage items per thread
    struct Bits { uint32_t v[kStage]; };
    auto load_bits = [&](int j0) -> Bits {
        Bits o;
#pragma unroll
        for (int e = 0; e < kStage; e++) {
            const int it = tid + e * 64 * NW, j = j0 + (it >> 3);
            o.v[e] = j < nt ? *(const uint32_t *)(tb + (int64_t)j * 32 + 4 * (it & 7)) : 0u;
        }
        return o;
    };
    auto store_tile = [&](uint8_t *dst, const Bits &in) {
#pragma unroll
        for (int e = 0; e < kStage; e++) {
            const int it = tid + e * 64 * NW, srow = it >> 3, sw = it & 7;
            const uint32_t bits = in.v[e];
            uint4 lo, hi;
            lo.x = spread_nibble(bits & 15u); lo.y = spread_nibble((bits >> 4) & 15u); lo.z = spread_nibble((bits >> 8) & 15u); lo.w = spread_nibble((bits >> 12) & 15u);
            hi.x = spread_nibble((bits >> 16) & 15u); hi.y = spread_nibble((bits >> 20) & 15u); hi.z = spread_nibble((bits >> 24) & 15u); hi.w = spread_nibble(bits >> 28);
            uint8_t *row = dst + srow * 256;
            *(uint4 *)(row + 16 * ((2 * sw) ^ (srow & 15))) = lo;                   // chunk 2 s + h of the row, XOR-swizzled
            *(uint4 *)(row + 16 * ((2 * sw + 1) ^ (srow & 15))) = hi;
        }
    };
    uint32_t best = 0;
    const int ntiles = (nt + kMmT - 1) / kMmT;
    // the packed bits of a tile are requested kMmAhead tiles before they are spread into LDS: a tile's products take
    // ~2 k cycles, a global load under load 2-5 k -- one tile ahead, every round of the workgroup waited for memory
    constexpr int kMmAhead = 4;
    Bits pre[kMmAhead];
#pragma unroll
    for (int d = 0; d < kMmAhead; d++) pre[d] = load_bits(d * kMmT);            // (rows beyond nt read as zero)
    if (ntiles > 0) { store_tile(tile[0], pre[0]); pre[0] = load_bits(kMmAhead * kMmT); }
    __syncthreads();
    for (int tl0 = 0; tl0 < ntiles; tl0 += kMmAhead) {
#pragma unroll
        for (int d = 0; d < kMmAhead; d++) {
            const int tl = tl0 + d;
            if (tl >= ntiles) break;                                            // uniform
            const uint8_t *cur = tile[d & 1];                                   // kMmAhead is even: tile tl lives in buffer tl & 1 = d & 1
#pragma unroll
            for (int m = 0; m < 2; m++) {                                       // the tile's two blocks of 32 train rows
                const int base = tl * kMmT + 32 * m;
                if (base >= nt) break;                                          // uniform
                v16i_t acc;
#pragma unroll
                for (int g = 0; g < 16; g++) acc[g] = 256;                      // dot + 256 >= 0
                const uint8_t *arow = cur + (32 * m + r) * 256;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const v4i_t a = *(const v4i_t *)(arow + 16 * ((2 * s + h) ^ (r & 15)));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, Bq[s], acc, 0, 0, 0);
                }
                uint32_t key[16];
#pragma unroll
                for (int g = 0; g < 16; g++) key[g] = ((uint32_t)acc[g] << 16) | Crow[g];
                if (nt - base < 32) {                                           // the last block: rows beyond nt never win
#pragma unroll
                    for (int g = 0; g < 16; g++) if ((int)(0xFFFFu - Crow[g]) >= nt - base) key[g] = 0u;
                }
                uint32_t mx = 0;
#pragma unroll
                for (int g = 0; g < 16; g++) mx = max(mx, key[g]);
                // the block's first row is train `base`: the low half becomes 0xFFFF - global index (no borrow: indices < 65536)
                best = max(best, mx > 0u ? mx - (uint32_t)base : 0u);
            }
            if (tl + 1 < ntiles) {
                // tile tl + 1 -> the other buffer (its readers passed the barrier below one tile ago), then its register is
                // free for tile tl + 1 + kMmAhead
                store_tile(tile[(d + 1) & 1], pre[(d + 1) % kMmAhead]);
                pre[(d + 1) % kMmAhead] = load_bits((tl + 1 + kMmAhead) * kMmT);
            }
            __syncthreads();
        }
    }
    best = max(best, (uint32_t)__shfl_xor((int)best, 32, 64));                      // the two lane halves hold different rows of the same query
    if (h == 0 && iq < nq) {
        const int dot = (int)(best >> 16) - 256;
        idx[(int64_t)b * out_stride + iq] = nt > 0 ? (int)(0xFFFFu - (best & 0xFFFFu)) : -1;
        dist[(int64_t)b * out_stride + iq] = nt > 0 ? (float)(popq - dot) : (float)(1 << 30);
    }
}

// Tracking::ORB_Robust_Find_MuliImage_MatchedFeatures' filter (src/tracking.cpp:546-577): one
// workgroup per pair; ordered emission of (t1_left, t1_right, t2_left).
struct OrbFilterArgs {
    const svo_keypoint *lastL, *lastR, *curL; int64_t kp_stride_prev, kp_stride_cur;   // per pair strides (keypoints)
    const int *nLastL, *nLastR, *nCurL; int n_stride;
    const int *i1, *i2; const float *d1, *d2; int64_t m_stride;
    double match_err;
    float2 *t1l, *t1r, *t2l; int64_t out_stride;
    int *m_out;
};
__global__ __launch_bounds__(256) void orb_filter_kernel(OrbFilterArgs a)
{
    __shared__ float s_min[4], s_max[4];
    __shared__ int s_base, s_wave[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nL = a.nLastL[b * a.n_stride], nR = a.nLastR[b * a.n_stride], nC = a.nCurL[b * a.n_stride];
    const int des_index = min(min(nL, nR), nC);
    const float *d1 = a.d1 + (int64_t)b * a.m_stride, *d2 = a.d2 + (int64_t)b * a.m_stride;
    const int *i1 = a.i1 + (int64_t)b * a.m_stride, *i2 = a.i2 + (int64_t)b * a.m_stride;
    float mn = 10000.f, mx = 0.f;
    for (int i = tid; i < des_index; i += 256) { const float d = fmaxf(d1[i], d2[i]); mn = fminf(mn, d); mx = fmaxf(mx, d); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, __shfl_xor(mn, m, 64)); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
    if (lane == 0) { s_min[wv] = mn; s_max[wv] = mx; }
    if (tid == 0) s_base = 0;
    __syncthreads();
    const double min_dist = (double)fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
    const double thr = 2 * min_dist > 30.0 ? 2 * min_dist : 30.0;
    const svo_keypoint *lastL = a.lastL + (int64_t)b * a.kp_stride_prev, *lastR = a.lastR + (int64_t)b * a.kp_stride_prev;
    const svo_keypoint *curL = a.curL + (int64_t)b * a.kp_stride_cur;
    const int64_t o = (int64_t)b * a.out_stride;
    for (int start = 0; start < des_index; start += 256) {
        const int i = start + tid;
        bool k = false;
        if (i < des_index)
            k = ((double)d1[i] <= thr) && ((double)d2[i] <= thr) &&
                ((double)fabsf(lastL[i].y - lastR[i1[i]].y) < a.match_err);
        const unsigned long long m = __ballot(k);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int q = 0; q < 4; q++) { if (q < wv) pre += s_wave[q]; tot += s_wave[q]; }
        if (k) {
            const int dst = s_base + pre + rank;
            a.t1l[o + dst] = make_float2(lastL[i].x, lastL[i].y);
            a.t1r[o + dst] = make_float2(lastR[i1[i]].x, lastR[i1[i]].y);
            a.t2l[o + dst] = make_float2(curL[i2[i]].x, curL[i2[i]].y);
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) a.m_out[b] = s_base;
}

// ---- host side --------------------------------------------------------------------------------
static int align_up_i(int v, int a) { return (v + a - 1) / a * a; }

// ORBextractor::ORBextractor + ComputePyramid geometry (all float/double expressions as the
// reference evaluates them)
int orb_make_geom(const svo_config &cfg, OrbGeom *g)
{
    memset(g, 0, sizeof(*g));
    const int nlevels = cfg.orb_nlevels, nfeatures = cfg.orb_nfeatures;
    if (nlevels < 1 || nlevels > kOrbMaxLevels) return SVO_ERR_ARG;
    g->nlevels = nlevels;
    const double scaleFactor = (double)cfg.orb_scale_factor;
    float inv_scale[kOrbMaxLevels];
    g->scale[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) g->scale[i] = (float)(g->scale[i - 1] * scaleFactor);
    for (int i = 0; i < nlevels; i++) inv_scale[i] = 1.0f / g->scale[i];
    const float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        g->quota[level] = (int)lrintf(nDesired);
        sum += g->quota[level];
        nDesired *= factor;
    }
    g->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    const int vmax = (int)floor(15 * sqrt(2.f) / 2 + 1), vmin = (int)ceil(15 * sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) g->umax[v] = (int)lrint(sqrt(225.0 - v * v));
    for (int v = 15, v0 = 0; v >= vmin; --v) {
        while (g->umax[v0] == g->umax[v0 + 1]) ++v0;
        g->umax[v] = v0;
        ++v0;
    }
    // getGaussianKernel(7, 2, CV_32F) * 256, rounded (legacy 8-bit fixed-point GaussianBlur)
    {
        double t[7], s = 0;
        for (int i = 0; i < 7; i++) { double x = i - 3; float f = (float)exp(-0.5 / (2.0 * 2.0) * x * x); t[i] = f; s += f; }
        s = 1. / s;
        for (int i = 0; i < 7; i++) { float f = (float)(t[i] * s); g->gk[i] = (int)lrintf(f * 256.f); }
    }
    int64_t off = 0, boff = 0;
    int coff = 0, boff_cells = 0;
    for (int l = 0; l < nlevels; l++) {
        g->w[l] = (int)lrintf((float)cfg.width * inv_scale[l]);
        g->h[l] = (int)lrintf((float)cfg.height * inv_scale[l]);
        if (g->w[l] < 1 || g->h[l] < 1) return SVO_ERR_ARG;
        g->pitch[l] = align_up_i(g->w[l] + 2 * kPad, 64);
        g->origin[l] = off + (int64_t)kPad * g->pitch[l] + kPad;
        off += (int64_t)g->pitch[l] * (g->h[l] + 2 * kPad);
        off = (off + 255) / 256 * 256;
        g->blur_off[l] = boff;
        g->bpitch[l] = (g->w[l] + 3) & ~3;
        boff += (int64_t)g->bpitch[l] * g->h[l];
        // cell grid of ComputeKeyPointsOctTree (:729-741)
        const int maxBX = g->w[l] - 16, maxBY = g->h[l] - 16;
        const float width = (float)(maxBX - 16), height = (float)(maxBY - 16);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        g->cell_off[l] = coff;
        if (nCols > 0 && nRows > 0 && width > 0 && height > 0) {
            g->nCols[l] = nCols; g->nRows[l] = nRows;
            g->wCell[l] = (int)ceil(width / nCols); g->hCell[l] = (int)ceil(height / nRows);
            if (g->wCell[l] + 6 > kCellMax || g->hCell[l] + 6 > kCellMax) return SVO_ERR_ARG;
            g->ncell[l] = nCols * nRows;
            if (g->ncell[l] > 1024) return SVO_ERR_ARG;
            int gc = (kCellPitch - 3 - 6) / g->wCell[l];                     // cells whose common window fits an LDS row
            gc = gc < 1 ? 1 : gc > kCellGroup ? kCellGroup : gc;
            g->gcell[l] = gc; g->gcols[l] = (nCols + gc - 1) / gc;
        }
        g->blk_off[l] = boff_cells;
        boff_cells += g->ncell[l] > 0 ? g->gcols[l] * g->nRows[l] : 0;
        coff += g->ncell[l];
    }
    g->slot_bytes = off;
    g->blur_total = boff;                       // (a multiple of 4: every image's levels start 4-byte aligned)
    g->cells_total = coff;
    g->blks_total = boff_cells;
    {
        int yb = 0;
        for (int l = 0; l < nlevels; l++) {
            g->blur_blk[l] = yb;
            yb += (g->h[l] + 4 * kBlurRowsPerThread - 1) / (4 * kBlurRowsPerThread);
        }
        for (int l = nlevels; l <= kOrbMaxLevels; l++) g->blur_blk[l] = yb;
    }
    int xo = 0, yo = 0;
    for (int l = 0; l < nlevels; l++) { g->xtab_off[l] = xo; g->ytab_off[l] = yo; if (l > 0) { xo += g->w[l]; yo += g->h[l]; } }
    g->xtab_total = xo; g->ytab_total = yo;
    return SVO_OK;
}

// cv::resize's coordinate / weight tables for every level l >= 1 (INTER_LINEAR, 11-bit weights;
// scale = 1 / (dst / src): CANONICAL O3).  x: clamp as upstream's xofs/alpha loop does, y: rows
// clamped to the source like upstream's vertical pass.
void orb_make_tables(const OrbGeom &g, std::vector<int2> &xt, std::vector<int4> &yt)
{
    xt.assign((size_t)g.xtab_total, make_int2(0, 0));
    yt.assign((size_t)g.ytab_total, make_int4(0, 0, 0, 0));
    for (int l = 1; l < g.nlevels; l++) {
        const int dw = g.w[l], dh = g.h[l], sw = g.w[l - 1], sh = g.h[l - 1];
        const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
        const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            const int a0 = (short)lrintf((1.f - fx) * 2048), a1 = (short)lrintf(fx * 2048);
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            xt[(size_t)g.xtab_off[l] + dx] = make_int2(sx | (sx1 << 16), (a0 & 0xFFFF) | (a1 << 16));
        }
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)floorf(fy);
            fy -= sy;
            const int b0 = (short)lrintf((1.f - fy) * 2048), b1 = (short)lrintf(fy * 2048);
            const int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
            const int y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
            yt[(size_t)g.ytab_off[l] + dy] = make_int4(y0, y1, b0, b1);
        }
    }
}

// Which levels the row-streaming resize kernel covers: every group of four output columns must find its taps inside the
// 8-byte window from its first source column, the weights must be non-negative 11-bit values, and the second source row of
// consecutive output rows must not decrease (the walk goes down the source rows once).  True for every scale factor in (1, 2].
void orb_resize_stream_levels(OrbGeom &g, const std::vector<int2> &xt, const std::vector<int4> &yt)
{
    for (int l = 0; l < kOrbMaxLevels; l++) g.rs_stream[l] = 0;
    for (int l = 1; l < g.nlevels; l++) {
        bool ok = true;
        const int dw = g.w[l], dh = g.h[l];
        for (int dx0 = 0; dx0 < dw && ok; dx0 += 4) {
            const int s0 = xt[(size_t)g.xtab_off[l] + dx0].x & 0xFFFF;
            for (int q = 0; q < 4; q++) {
                const int2 t = xt[(size_t)g.xtab_off[l] + (dx0 + q < dw ? dx0 + q : dw - 1)];
                const int a0 = (short)(t.y & 0xFFFF), a1 = t.y >> 16;
                if ((t.x & 0xFFFF) < s0 || (t.x >> 16) - s0 > 7 || (t.x >> 16) < (t.x & 0xFFFF) || a0 < 0 || a1 < 0 || a0 > 2048 || a1 > 2048) ok = false;
            }
        }
        for (int dy = 0; dy < dh && ok; dy++) {
            const int4 t = yt[(size_t)g.ytab_off[l] + dy];
            if (t.y < t.x || t.y > t.x + 1 || t.z < 0 || t.w < 0 || t.z > 2048 || t.w > 2048) ok = false;
            if (dy > 0 && t.y < yt[(size_t)g.ytab_off[l] + dy - 1].y) ok = false;
        }
        g.rs_stream[l] = ok ? 1 : 0;
    }
}


int orb_alloc(svo_ctx *ctx)
{
    if (ctx->orb_ready) return SVO_OK;
    if (ctx->cfg.max_keypoints > 16384) {          // 16-bit indices: matcher keys, quadtree candidate indices (4 x max_keypoints)
        ctx->err = "ORB mode: max_keypoints must be <= 16384";
        return SVO_ERR_ARG;
    }
    OrbGeom &g = ctx->orb_geom;
    int rc = orb_make_geom(ctx->cfg, &g);
    if (rc) { ctx->err = "ORB geometry: unsupported image size / level count"; return rc; }
    // DistributeOctTree starts from nIni = round(width / height) root strips of the level's keypoint area; both quadtree
    // kernels hold at most 64 of them.  A panorama beyond 64 : 1 is refused here instead of losing the strips silently.
    for (int l = 0; l < g.nlevels; l++) {
        const int aw = g.w[l] - 32, ah = g.h[l] - 32;
        if (aw > 0 && ah > 0 && (int)roundf((float)aw / (float)ah) > 64) {
            ctx->err = "ORB mode: a pyramid level is wider than 64 : 1 (more than 64 quadtree root strips): unsupported";
            return SVO_ERR_ARG;
        }
    }
    const int n_img = 2 * ctx->n_img;                  // left + right of every frame slot
    const int L = g.nlevels;
    {
        // quadtree LDS: room for the largest per-level quota; opt in to more than the default dynamic limit
        int qmax = 0;
        for (int l = 0; l < g.nlevels; l++) qmax = qmax > g.quota[l] ? qmax : g.quota[l];
        ctx->orb_node_cap = (qmax + 16 + 63) / 64 * 64;
        const size_t bytes = qlds_bytes(ctx->orb_node_cap);
        if (bytes > 150 * 1024 || ctx->orb_node_cap > 4096) {
            ctx->err = "ORB: nFeatures per level too large for the quadtree's LDS";
            return SVO_ERR_ARG;
        }
        SVO_HIP(hipFuncSetAttribute((const void *)orb_distribute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        // the node-parallel kernel: positions are 12-bit fields of the expansion keys, candidates must fit its LDS
        ctx->orb_qt_parallel = qplds_bytes(ctx->orb_node_cap) <= 150 * 1024 && ctx->orb_node_cap <= 4096 && getenv("SVO_ORB_QT_SERIAL") == nullptr;
        if (ctx->orb_qt_parallel)
            SVO_HIP(hipFuncSetAttribute((const void *)orb_distribute_par_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)qplds_bytes(ctx->orb_node_cap)));
    }
    // FAST candidates kept per (image, level); one less than 2^16 at most: a quadtree node's key count is a 16-bit field
    ctx->orb_cand_cap = 4 * ctx->cfg.max_keypoints < 65535 ? 4 * ctx->cfg.max_keypoints : 65535;
    const int kCandCap = ctx->orb_cand_cap;
    SVO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), svo_bit_pattern_31, 1024));
#define DA(ptr, bytes) do { if (dev_alloc(ctx, &(ptr), (bytes)) != SVO_OK) return SVO_ERR_HIP; } while (0)
    {
        std::vector<int2> xt; std::vector<int4> yt;
        orb_make_tables(g, xt, yt);
        orb_resize_stream_levels(g, xt, yt);
        ctx->orb_resize_staged = getenv("SVO_ORB_RESIZE_STAGED") != nullptr;      // the round-4 kernel, for A/B measurements and its test
        ctx->orb_copy_level0 = getenv("SVO_ORB_COPY_LEVEL0") != nullptr;          // level 0 always copied into the slot (A/B, its test)
        DA(ctx->orb_xtab, sizeof(int2) * (xt.size() + 1));
        DA(ctx->orb_ytab, sizeof(int4) * (yt.size() + 1));
        // (inside svo_create the buffers exist only after the context's one allocation: the uploads wait for it)
        const int rc_up = dev_defer(ctx, [ctx, xt, yt]() {
            if (!xt.empty()) SVO_HIP(hipMemcpy(ctx->orb_xtab, xt.data(), sizeof(int2) * xt.size(), hipMemcpyHostToDevice));
            if (!yt.empty()) SVO_HIP(hipMemcpy(ctx->orb_ytab, yt.data(), sizeof(int4) * yt.size(), hipMemcpyHostToDevice));
            return SVO_OK;
        });
        if (rc_up != SVO_OK) return rc_up;
    }
    DA(ctx->orb_slots, (size_t)g.slot_bytes * n_img);
    DA(ctx->orb_blur, (size_t)g.blur_total * n_img + 64);     // + the descriptor kernel's last dword past a row end
    DA(ctx->orb_cell_cand, sizeof(float4) * (size_t)g.cells_total * kCellCap * n_img);
    DA(ctx->orb_cell_cnt, sizeof(int) * (size_t)g.cells_total * n_img);
    DA(ctx->orb_lvl_cand, sizeof(float4) * (size_t)kCandCap * L * n_img);
    DA(ctx->orb_lvl_cnt, sizeof(int) * (size_t)L * n_img);
    DA(ctx->orb_qkeys, sizeof(uint2) * (size_t)kCandCap * L * n_img);
    DA(ctx->orb_qtmp, sizeof(uint2) * (size_t)kCandCap * L * n_img);
    DA(ctx->orb_sel, sizeof(int) * (size_t)ctx->orb_node_cap * L * n_img);
    DA(ctx->orb_sel_cnt, sizeof(int) * (size_t)L * n_img);
    DA(ctx->orb_overflow, sizeof(int) * (size_t)n_img);
    {
        const int rc_z = dev_defer(ctx, [ctx, n_img]() { SVO_HIP(hipMemset(ctx->orb_overflow, 0, sizeof(int) * (size_t)n_img)); return SVO_OK; });
        if (rc_z != SVO_OK) return rc_z;
    }
    ctx->orb_kp_cap = ctx->cfg.max_keypoints;
    DA(ctx->orb_kps, sizeof(svo_keypoint) * (size_t)ctx->orb_kp_cap * n_img);
    DA(ctx->orb_desc, (size_t)32 * ctx->orb_kp_cap * n_img);
    DA(ctx->orb_n, sizeof(int) * (size_t)n_img);
    const int B = ctx->cfg.max_batch;
    for (int k = 0; k < 2; k++) {
        DA(ctx->orb_midx[k], sizeof(int) * (size_t)ctx->orb_kp_cap * B);
        DA(ctx->orb_mdist[k], sizeof(float) * (size_t)ctx->orb_kp_cap * B);
    }
#undef DA
    ctx->orb_ready = true;
    return SVO_OK;
}

void orb_free(svo_ctx *) {}        // the ORB buffers belong to the context's arena (or its lazy extras): freed with it

// ORBextractor::operator() on `n_img` images (image b at img + b*img_stride, or interleaved L/R
// when img2 != null) into output slots [slot0, slot0 + n_img).
int orb_extract_batch(svo_ctx *ctx, const uint8_t *img, const uint8_t *img2, int pitch, int64_t img_stride, int slot0,
                      int n_img, hipStream_t st, bool in_place)
{
    const OrbGeom &g = ctx->orb_geom;
    const int L = g.nlevels, kCandCap = ctx->orb_cand_cap;
    uint8_t *slots = ctx->orb_slots + (size_t)slot0 * g.slot_bytes;
    dim3 blk(256);
    int *ovf = ctx->orb_overflow + slot0;
    SVO_HIP(hipMemsetAsync(ovf, 0, sizeof(int) * (size_t)n_img, st));
    // Level 0 is the input image: read IN PLACE when the caller says the images stay put until this call's kernels are done
    // (the batched and online paths: frame buffers / staging owned by the context) and they can be read the way the slots are --
    // 16-byte aligned rows with at least 16 bytes of padding (the blur, the descriptor windows and the aligned staging loads
    // run a few bytes past a row's last pixel).  Otherwise (stage API, odd pitches) it is copied into the slot first:
    // 0.16 ms per 514 images alone, 0.36 ms beside the previous batch's pose stage.
    OrbL0 z{};
    if (in_place && !ctx->orb_copy_level0 && ((uintptr_t)img & 15) == 0 && (!img2 || ((uintptr_t)img2 & 15) == 0) && (pitch & 15) == 0 && (img_stride & 15) == 0 &&
        pitch >= g.w[0] + 16)
        z = OrbL0{img, img2, pitch, img_stride};
    ctx->orb_level0_in_slot = z.img == nullptr;
    {
        const int n = n_img;
        uint8_t *sl = slots;
        if (!z.img) {
            const int lanes = (g.w[0] + 15) / 16, bx = lanes < 256 ? lanes : 256, by = 256 / bx;
            const dim3 cblk(bx, by), cgrid((lanes + bx - 1) / bx, (g.h[0] + by - 1) / by, img2 ? n / 2 : n);
            if (img2) {
                // left images -> even slots, right images -> odd slots
                hipLaunchKernelGGL(orb_copy0_kernel, cgrid, cblk, 0, st, g, img, pitch, img_stride, sl, 2 * g.slot_bytes);
                hipLaunchKernelGGL(orb_copy0_kernel, cgrid, cblk, 0, st, g, img2, pitch, img_stride, sl + g.slot_bytes, 2 * g.slot_bytes);
            } else {
                hipLaunchKernelGGL(orb_copy0_kernel, cgrid, cblk, 0, st, g, img, pitch, img_stride, sl, g.slot_bytes);
            }
        }
        for (int l = 1; l < L; l++) {
            if (g.rs_stream[l] && !ctx->orb_resize_staged) {
                // output rows per wave: long bands amortise a wave's set-up (its table entries, the first source row), short ones
                // keep >= 16 K waves in the launch (the small levels, the online pair)
                const int gx = ((g.w[l] + 3) / 4 + 63) / 64;
                int band = (int)((int64_t)g.h[l] * gx * n / 16384) & ~7;
                band = band < 8 ? 8 : band > 32 ? 32 : band;
                const int gy = (g.h[l] + 4 * band - 1) / (4 * band);
                hipLaunchKernelGGL(orb_resize_stream_kernel, dim3(gx * gy * n), blk, 0, st, g, sl, g.slot_bytes, l,
                                   (const int2 *)ctx->orb_xtab, (const int4 *)ctx->orb_ytab, n, gx, gy, band, z);
                continue;
            }
            // source rows a strip of kResizeRows output rows can touch (+3: second tap, rounding), and the
            // dwords 1024 output columns can span (+3: second tap, alignment slack)
            int cap_rows = (int)((double)kResizeRows * g.h[l - 1] / g.h[l]) + 3;
            int row_dw = (((int)(1024.0 * g.w[l - 1] / g.w[l] / 4.0) + 3 + 4) + 3) & ~3;     // + the 16-byte alignment slack, multiple of 4
            if ((size_t)cap_rows * row_dw * 4 > 60 * 1024) cap_rows = 60 * 1024 / (row_dw * 4);     // beyond it: the unstaged path
            hipLaunchKernelGGL(orb_resize_kernel, dim3((g.w[l] + 1023) / 1024, (g.h[l] + kResizeRows - 1) / kResizeRows, n), blk,
                               (size_t)cap_rows * row_dw * 4, st, g, sl, g.slot_bytes, l,
                               (const int2 *)ctx->orb_xtab, (const int4 *)ctx->orb_ytab, cap_rows, row_dw, z);
        }
        // the blurred levels only depend on the pyramid: computed here, before the LDS-hungry kernels,
        // so that in overlap mode the previous batch's pose solver (74 KB of LDS per workgroup) runs
        // beside kernels that need no LDS
        uint8_t *blur = ctx->orb_blur + (size_t)slot0 * g.blur_total;
        hipLaunchKernelGGL(orb_blur_kernel, dim3((g.w[0] + 255) / 256, g.blur_blk[L], n), dim3(64, 4), 0, st, g, sl, g.slot_bytes,
                           blur, g.blur_total, z);
    }
    timing_mark(ctx, "orb_pyramid");
    float4 *cell_cand = ctx->orb_cell_cand + (size_t)slot0 * g.cells_total * kCellCap;
    int *cell_cnt = ctx->orb_cell_cnt + (size_t)slot0 * g.cells_total;
    {
        int hmax = 0;
        for (int l = 0; l < L; l++) if (g.ncell[l] > 0 && g.hCell[l] > hmax) hmax = g.hCell[l];
        // The dynamic LDS of a launch is sized by its tallest cells, and the size sets how many workgroups a CU holds (the
        // kernel is issue-bound: 5 -> 6 workgroups per CU took 12 % off it).  A batch therefore launches runs of consecutive levels
        // that reach the same number of workgroups per CU separately -- KITTI: levels 0-3 and 6 (cells of 31-33 rows, 22 KB) at
        // seven per CU, levels 4, 5, 7 (37-40 rows, 26 KB) at six --; a launch of a few images stays one launch (latency).
        // (SVO_ORB_CF_LDS5=1: the round-5 allocation of five planes -- same carve-up, five workgroups per CU --, =2: one launch
        // sized by the tallest cells, for A/B runs)
        static const int lds_mode = getenv("SVO_ORB_CF_LDS5") ? atoi(getenv("SVO_ORB_CF_LDS5")) : 0;
        auto per_cu = [](size_t bytes) { return (int)((size_t)160 * 1024 / (((bytes + 64) + 1023) & ~(size_t)1023)); };
        const size_t cf_all = lds_mode == 1 ? (size_t)5 * cellfast_plane(hmax) + 16 : cellfast_lds_bytes(hmax);
        if (g.blks_total > 0 && (lds_mode != 0 || n_img < 16)) {
            hipLaunchKernelGGL(orb_cellfast_kernel, dim3(g.blks_total * n_img), dim3(kCellThreads), cf_all, st,
                               g, slots, g.slot_bytes, ctx->cfg.orb_ini_th, ctx->cfg.orb_min_th, cell_cand, cell_cnt,
                               (int64_t)g.cells_total * kCellCap, (int64_t)g.cells_total, n_img, z, 0, g.blks_total);
        } else if (g.blks_total > 0) {
            for (int l0 = 0; l0 < L;) {
                int l1 = l0, hrun = 0;
                const int cls = per_cu(cellfast_lds_bytes(g.hCell[l0]));
                while (l1 < L && (g.ncell[l1] == 0 || per_cu(cellfast_lds_bytes(g.hCell[l1])) == cls)) {
                    if (g.ncell[l1] > 0 && g.hCell[l1] > hrun) hrun = g.hCell[l1];
                    l1++;
                }
                const int lo = g.blk_off[l0], hi = l1 < L ? g.blk_off[l1] : g.blks_total;
                if (hi > lo)
                    hipLaunchKernelGGL(orb_cellfast_kernel, dim3((hi - lo) * n_img), dim3(kCellThreads), cellfast_lds_bytes(hrun), st,
                                       g, slots, g.slot_bytes, ctx->cfg.orb_ini_th, ctx->cfg.orb_min_th, cell_cand, cell_cnt,
                                       (int64_t)g.cells_total * kCellCap, (int64_t)g.cells_total, n_img, z, lo, hi - lo);
                l0 = l1;
            }
        }
    }
    float4 *lvl_cand = ctx->orb_lvl_cand + (size_t)slot0 * L * kCandCap;
    int *lvl_cnt = ctx->orb_lvl_cnt + (size_t)slot0 * L;
    hipLaunchKernelGGL(orb_gather_kernel, dim3(L, n_img), blk, 0, st, g, cell_cand, cell_cnt, (int64_t)g.cells_total * kCellCap,
                       (int64_t)g.cells_total, lvl_cand, lvl_cnt, kCandCap, ovf);
    timing_mark(ctx, "orb_cellfast");
    OrbDistArgs d{};
    d.g = g; d.lvl_cand = lvl_cand; d.lvl_cnt = lvl_cnt; d.cand_cap = kCandCap;
    d.gkeys = (uint2 *)ctx->orb_qkeys + (size_t)slot0 * L * kCandCap; d.gtmp = (uint2 *)ctx->orb_qtmp + (size_t)slot0 * L * kCandCap;
    d.sel = ctx->orb_sel + (size_t)slot0 * L * ctx->orb_node_cap; d.sel_cnt = ctx->orb_sel_cnt + (size_t)slot0 * L; d.sel_cap = ctx->orb_node_cap;
    d.overflow = ovf;
    d.node_cap = ctx->orb_node_cap;
    if (ctx->orb_qt_parallel)
        hipLaunchKernelGGL(orb_distribute_par_kernel, dim3(n_img, L), dim3(kQpThreads), qplds_bytes(ctx->orb_node_cap), st, d);
    else
        hipLaunchKernelGGL(orb_distribute_kernel, dim3(n_img, L), dim3(64), qlds_bytes(ctx->orb_node_cap), st, d);
    timing_mark(ctx, "orb_quadtree");
    OrbDescArgs e{};
    e.g = g; e.z = z; e.slots = slots; e.slot_stride = g.slot_bytes; e.blur = ctx->orb_blur + (size_t)slot0 * g.blur_total; e.blur_img_stride = g.blur_total;
    e.lvl_cand = lvl_cand; e.cand_cap = kCandCap; e.sel = d.sel; e.sel_cnt = d.sel_cnt; e.sel_cap = ctx->orb_node_cap;
    e.kps = (svo_keypoint *)ctx->orb_kps + (size_t)slot0 * ctx->orb_kp_cap; e.desc = ctx->orb_desc + (size_t)slot0 * ctx->orb_kp_cap * 32;
    e.n_out = ctx->orb_n + slot0; e.out_cap = ctx->orb_kp_cap; e.overflow = ovf;
    int max_kp = 0;
    for (int l = 0; l < L; l++) max_kp += g.quota[l] + 8;
    if (max_kp > ctx->orb_kp_cap) max_kp = ctx->orb_kp_cap;
    e.blocks_per_img = (max_kp + 7) / 8; e.n_img = n_img;
    hipLaunchKernelGGL(orb_describe_kernel, dim3(e.blocks_per_img * n_img), blk, 0, st, e);    // two keypoints per wave
    timing_mark(ctx, "orb_describe");
    return SVO_OK;
}

void orb_launch_match_fixed(svo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, hipStream_t st)
{
    hipLaunchKernelGGL(orb_match_kernel<4>, dim3((nq + 127) / 128, 1), dim3(256), 0, st, q, (const int *)nullptr, nq, (int64_t)0, t,
                       (const int *)nullptr, nt, (int64_t)0, 0, ctx->orb_midx[0], ctx->orb_mdist[0], (int64_t)ctx->orb_kp_cap);
}

// match + filter for n_pairs pairs: pair p = (frame fp0 + p*fstep, frame fc0 + p*fstep); writes
// cmp[0] = t1_left, cmp[1] = t1_right, cmp[3] = t2_left, m_out
int orb_match_pairs(svo_ctx *ctx, int n_pairs, int fp0, int fc0, int fstep, hipStream_t st)
{
    const int cap = ctx->orb_kp_cap;
    const uint8_t *D = ctx->orb_desc;
    const int *N = ctx->orb_n;
    const svo_keypoint *K = (const svo_keypoint *)ctx->orb_kps;
    const int64_t fs = (int64_t)fstep * 2;                         // image slots per pair step
    // an image never holds more keypoints than the per-level quotas allow (+ the extractor's slack): the grid is sized
    // by that, not by the capacity of the buffers (three quarters of the workgroups were launched to return at once)
    int max_kp = 0;
    for (int l = 0; l < ctx->orb_geom.nlevels; l++) max_kp += ctx->orb_geom.quota[l] + 8;
    if (max_kp > cap) max_kp = cap;
    const bool small = (int64_t)n_pairs * ((max_kp + 255) / 256) < 256;              // fewer workgroups than CUs: latency shape
    const int qblocks = small ? (max_kp + 127) / 128 : (max_kp + 255) / 256, qthreads = small ? 256 : 512;
    auto kern = small ? orb_match_kernel<4> : orb_match_kernel<8>;
    // match1: last.left -> last.right ; match2: last.left -> cur.left  (src/tracking.cpp:543-544)
    hipLaunchKernelGGL(kern, dim3(qblocks, n_pairs), dim3(qthreads), 0, st, D + (size_t)(2 * fp0) * cap * 32,
                       N + 2 * fp0, 0, fs * cap * 32, D + (size_t)(2 * fp0 + 1) * cap * 32, N + 2 * fp0 + 1, 0, fs * cap * 32,
                       (int)fs, ctx->orb_midx[0], ctx->orb_mdist[0], (int64_t)cap);
    hipLaunchKernelGGL(kern, dim3(qblocks, n_pairs), dim3(qthreads), 0, st, D + (size_t)(2 * fp0) * cap * 32,
                       N + 2 * fp0, 0, fs * cap * 32, D + (size_t)(2 * fc0) * cap * 32, N + 2 * fc0, 0, fs * cap * 32,
                       (int)fs, ctx->orb_midx[1], ctx->orb_mdist[1], (int64_t)cap);
    OrbFilterArgs f{};
    f.lastL = K + (size_t)(2 * fp0) * cap; f.lastR = K + (size_t)(2 * fp0 + 1) * cap; f.curL = K + (size_t)(2 * fc0) * cap;
    f.kp_stride_prev = fs * cap; f.kp_stride_cur = fs * cap;
    f.nLastL = N + 2 * fp0; f.nLastR = N + 2 * fp0 + 1; f.nCurL = N + 2 * fc0; f.n_stride = (int)fs;
    f.i1 = ctx->orb_midx[0]; f.i2 = ctx->orb_midx[1]; f.d1 = ctx->orb_mdist[0]; f.d2 = ctx->orb_mdist[1]; f.m_stride = cap;
    f.match_err = ctx->cfg.feature_match_error;
    f.t1l = ctx->cmp[0]; f.t1r = ctx->cmp[1]; f.t2l = ctx->cmp[3]; f.out_stride = ctx->cfg.max_keypoints;
    f.m_out = ctx->m_out;
    hipLaunchKernelGGL(orb_filter_kernel, dim3(n_pairs), dim3(256), 0, st, f);
    return SVO_OK;
}

}  // namespace svo
