// lk.hip -- pyramidal iterative Lucas-Kanade tracking for gfx950: the body of
// cv::calcOpticalFlowPyrLK(prev, next, pts, out, status, err, Size(21,21), 3,
//                          TermCriteria(COUNT+EPS, 30, 0.01), 0, 0.001)
// as called four times per frame by Tracking::LK_Robust_Find_MuliImage_MatchedFeatures
// (reference src/tracking.cpp:583-622), plus the deleteBadmatchFeatures predicate (:623-660).
//
// Mapping: ONE WAVEFRONT TRACKS FOUR POINTS ("slots"), four waves per workgroup, no workgroup
// barrier.  The kernel is VALU-issue bound (tiles come from L2), and more than half of a
// one-point-per-wave iteration is per-point SCALAR work (window position, bilinear weights, 2x2
// solve, convergence tests) that a wave executes as full vector instructions.  Here that scalar work
// is done once for four points: lane l carries the control state of slot l >> 4 (its row of 16
// lanes), so one vector instruction advances all four points, while the pixel work is done slot
// after slot by all 63 pixel lanes:
//   * lane l owns window row l/3, columns (l%3)*7..+6 of EVERY slot (63 lanes x 7 px = 441 px);
//   * tiles live in LDS as ROW-PAIR COLUMN WORDS (pixel[r][c] | pixel[r+1][c] << 16), formed once while
//     staging (4 v_perm per source dword pair): exactly the operand of the bilinear v_dot2_i32_i16, so
//     neither the patch build nor an iteration spends an instruction on byte alignment or unpacking;
//   * per level, per slot: the 24x28 I tile (requested one level ahead); the lane reads its 3 row pairs x
//     10 columns, forms the Scharr derivatives of the 2x8 positions it interpolates from ON THE FLY with
//     packed 16-bit math (the reference path materialises a full int16x2 derivative image per level
//     per call), interpolates I, Ix, Iy with v_dot2_i32_i16 and keeps the patch as 12 packed VGPRs;
//   * per iteration, per active slot: eight conflict-free dword reads from a column-major 27x28 J tile
//     (re-staged only when the window drifts out of it; requested beside the patch arithmetic), a
//     four-tap bilinear sample is two v_dot2_i32_i16 (signed 16-bit weight pairs; the words hold
//     pixel << 7, so the sample is the high half of its sum), then v_dot2 mismatch sums;
//   * the 16 partial sums of an iteration (4 slots x {b1,b2} x {low,high half}) are reduced with ONE
//     reduce-scatter over the rows (v_permlane32/16_swap + add, then DPP inside the row) that leaves
//     slot s's four sums in every quad of row s, exactly where that slot's control lanes need them.
// With ncalls == 4 the wave walks the whole circular chain L1 -> R1 -> R2 -> L2 -> L1' for its
// four points and stops early once all of them are rejected.  The column-word J tiles cost 4x the
// LDS of byte tiles (52 KB per workgroup): three waves per SIMD, 168 VGPRs.
//
// Exactness: all pixel arithmetic is upstream's fixed point (14-bit weights, 5 fractional bits);
// the five sums A11,A12,A22,b1,b2 are accumulated as exact integers (per-lane int32 partials,
// 16-bit halves reduced separately, recombined with one fused multiply-add) and converted to float
// once -- the canonical recipe of oracle/lk.c, so status bytes and point coordinates are
// bit-identical to the oracle.  FP contraction is off.
//
// Algorithmic HBM bytes (SURVEY.md 8d gather convention): per point per call
//   sum over 4 levels (24*24 + 22*22) + 8 in + 8 out + 1 status = 4257 B.
#include <cstring>
#include <type_traits>
#include "lk_common.h"

namespace svo {

// Diagnostic build (-DSVO_LK_STAMP=k, tools/gpu/lk_stamps.sh): every wave adds the s_memtime cycles it spends in
// section k of lk_call4 (between stamp points k and k + 1; k = 8: the whole call) to a counter read back with
// svo_debug_lk_stamps.  ONE section per build: a stamp drains the LDS queue and waits for the scalar memory
// unit, and stamping every section at once tripled the kernel's time and measured mostly the stamps.
#ifdef SVO_LK_STAMP
__device__ unsigned long long g_lk_stamp[2 * 1024];      // 1024 slots (by workgroup) against atomic contention; summed on read-back
__device__ __forceinline__ uint32_t lk_now() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)__builtin_amdgcn_s_memtime()); }
#define LK_STAMP_DECL uint32_t lk_t_ = 0, lk_acc_ = 0, lk_n_ = 0
#define LK_AT(i) do { if ((i) == SVO_LK_STAMP) lk_t_ = lk_now(); else if ((i) == SVO_LK_STAMP + 1) { lk_acc_ += lk_now() - lk_t_; lk_n_++; } } while (0)
#define LK_CALL_BEGIN do { if (SVO_LK_STAMP == 8) lk_t_ = lk_now(); } while (0)
#define LK_CALL_END(lane) do { if (SVO_LK_STAMP == 8) { lk_acc_ += lk_now() - lk_t_; lk_n_++; } \
    if ((lane) == 0) { atomicAdd(&g_lk_stamp[2 * (blockIdx.x & 1023)], (unsigned long long)lk_acc_); atomicAdd(&g_lk_stamp[2 * (blockIdx.x & 1023) + 1], (unsigned long long)lk_n_); } } while (0)
#else
#define LK_STAMP_DECL
#define LK_AT(i)
#define LK_CALL_BEGIN
#define LK_CALL_END(lane)
#endif


// per-lane constants of the pixel role
struct PixLane { int row, seg; uint32_t onmask, qoff; };   // qoff: byte offset of the lane's first I-tile word inside a slot tile

// ---- phase A for one slot: this lane's 7 patch pixels from the staged I tile -------------------
// Tile rows row..row+3 = image rows ipy+row-1..ipy+row+2, bytes j = 0..9 = image columns
// ipx-1+seg*7+j.  Everything is packed as COLUMN WORDS pairing two vertically adjacent rows
// (low half = upper row): Q01/Q12/Q23[j] = rows (0,1)/(1,2)/(2,3) of tile column j.  The packed
// Scharr passes then produce the derivative rows A (image row ipy+row) and B (ipy+row+1) side by
// side in one register per column, which is exactly the operand the bilinear v_dot2 wants:
//   val_k = dot2(D[k], (w00 | w10 << 16)) + dot2(D[k+1], (w01 | w11 << 16)) + rounding
// so no lane ever realigns a pixel pair.
// Outputs: packed patch derivatives (Ix, Iy as 4 pairs each; pair 3 has a zero high half), the
// lane's NEGATED constants -sum(I*Ix), -sum(I*Iy) over its 7 pixels, and the three partial sums of
// Ix^2, Ix*Iy, Iy^2.  The iterations need I only inside sum((J - I) * Ix) = sum(J*Ix) - sum(I*Ix):
// the second term does not change, so the per-iteration mismatch chain starts from the negated
// constant instead of subtracting I from every J sample (exact: integers, |partial| < 2^29).
template <bool EDGE>
__device__ __forceinline__ void patch_slot(uint32_t tile_addr, const PixLane &pl, uint32_t Wau,
                                           uint32_t Wbu, int ipx, int ipy, int w, int h,
                                           uint32_t (&IxP)[4], uint32_t (&IyP)[4], int &nIIx, int &nIIy,
                                           int &pA11, int &pA12, int &pA22)
{
    uint32_t IvP[4];
    // (lane 63 carries no window pixel: zero weights make its I, Ix, Iy and sums vanish; `onmask` is
    //  all ones in the pixel lanes, zero in lane 63 -- one v_and per weight word)
    const uint32_t Wa = Wau & pl.onmask, Wb = Wbu & pl.onmask;
    uint32_t Q01[10], Q12[10], Q23[10];
    {
        // ONE address per slot (tile_addr: the slot's tile + offI columns, wave-uniform; qoff: the lane's
        // column / row-pair offset): everything else is an immediate offset of the reads
        lds_cu32 *q0 = (lds_cu32 *)(size_t)(tile_addr + pl.qoff);
#pragma unroll
        for (int j = 0; j < 10; j++) { Q01[j] = q0[j * kQColDw]; Q12[j] = q0[j * kQColDw + 1]; Q23[j] = q0[j * kQColDw + 2]; }
    }
    // vertical Scharr passes, rows A | B packed.  Both passes carry a factor 4 (coefficients 12 / 40
    // instead of 3 / 10; |4 d| <= 16320 still fits 16 bits): the interpolated derivative
    // (sum + 2^13) >> 14 then equals (4 sum + 2^15) >> 16, i.e. the HIGH half of the accumulator, and
    // the pack below picks bytes 2-3 directly instead of shifting every value first.
    uint32_t T0[10], T1[10];
    const u16x2 k12 = {12, 12}, k40 = {40, 40};
#pragma unroll
    for (int j = 0; j < 10; j++) {
        T0[j] = as_u32((as_u16x2(Q01[j]) + as_u16x2(Q23[j])) * k12 + as_u16x2(Q12[j]) * k40);       // 4 t0
        T1[j] = as_u32(as_u16x2(Q23[j]) - as_u16x2(Q01[j]));                                          // t1
    }
    // horizontal passes: derivative column c = 0..7 (image column ipx+seg*7+c) from tile columns c..c+2
    uint32_t DX[8], DY[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        DX[c] = as_u32(as_u16x2(T0[c + 2]) - as_u16x2(T0[c]));                                        // 4 dx
        DY[c] = as_u32((as_u16x2(T1[c]) + as_u16x2(T1[c + 2])) * k12 + as_u16x2(T1[c + 1]) * k40);    // 4 dy
    }
    // the derivative image's border is BORDER_CONSTANT 0: mask positions outside the image
    // (only possible when the window hangs over the edge: the EDGE instantiation)
    if (EDGE) {
        const int gyA = ipy + pl.row, gyB = gyA + 1;
        const uint32_t rows = ((gyA >= 0 && gyA < h) ? 0x0000FFFFu : 0u) | ((gyB >= 0 && gyB < h) ? 0xFFFF0000u : 0u);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int gx = ipx + pl.seg * 7 + c;
            const uint32_t mk = (gx >= 0 && gx < w) ? rows : 0u;
            DX[c] &= mk; DY[c] &= mk;
        }
    }
    int iv[8], ix[8], iy[8];                                    // ix, iy: value << 16 | rounding residue
    iv[7] = ix[7] = iy[7] = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        iv[k] = dot2(Q12[k + 2], Wb, dot2_k(Q12[k + 1], Wa, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        ix[k] = dot2(DX[k + 1], Wb, dot2_k(DX[k], Wa, 1 << (W_BITS + 1)));
        iy[k] = dot2(DY[k + 1], Wb, dot2_k(DY[k], Wa, 1 << (W_BITS + 1)));
    }
    pA11 = 0; pA12 = 0; pA22 = 0;
    int sIIx = 0, sIIy = 0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        IvP[m] = perm_b32((uint32_t)iv[2 * m + 1], (uint32_t)iv[2 * m], 0x05040100u);
        IxP[m] = perm_b32((uint32_t)ix[2 * m + 1], (uint32_t)ix[2 * m], 0x07060302u);      // the two high halves
        IyP[m] = perm_b32((uint32_t)iy[2 * m + 1], (uint32_t)iy[2 * m], 0x07060302u);
        pA11 = m == 0 ? dot2_0(IxP[m], IxP[m]) : dot2(IxP[m], IxP[m], pA11);
        pA12 = m == 0 ? dot2_0(IxP[m], IyP[m]) : dot2(IxP[m], IyP[m], pA12);
        pA22 = m == 0 ? dot2_0(IyP[m], IyP[m]) : dot2(IyP[m], IyP[m], pA22);
        sIIx = m == 0 ? dot2_0(IvP[m], IxP[m]) : dot2(IvP[m], IxP[m], sIIx);
        sIIy = m == 0 ? dot2_0(IvP[m], IyP[m]) : dot2(IvP[m], IyP[m], sIIy);
    }
    nIIx = -sIIx; nIIy = -sIIy;
}

// `off` = byte offset of the lane's first J sample inside the slot's tile: (cy + row) * 40 + cx + seg * 7
// (the slot part is one value broadcast from its control lane, the lane part a constant)
// `vround` = 2^15 (the J rounding 2^8, scaled like the column words) held in a VGPR: a VOP3P instruction can read ONE scalar operand, and that one is the
// slot's weight (an SGPR from v_readlane); a scalar rounding constant cost a v_mov per slot and iteration
__device__ __forceinline__ void mismatch_slot(const uint32_t (&C)[8], uint32_t Wa, uint32_t Wb,
                                              const uint32_t (&IxP)[4], const uint32_t (&IyP)[4], int nIIx, int nIIy,
                                              int vround, int &pb1, int &pb2)
{
    int d[7];
#pragma unroll
    for (int k = 0; k < 7; k++) d[k] = dot2(C[k + 1], Wb, dot2_sv(C[k], Wa, vround));
    // J samples = high halves of the sums (column words hold pixel << 7, vround = 2^15: (sum * 2^7) >> 16
    // = sum >> 9, 0 <= sum < 2^22): two of them side by side with one v_perm
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const uint32_t vp = m < 3 ? perm_b32((uint32_t)d[2 * m + 1], (uint32_t)d[2 * m], 0x07060302u)
                                  : (uint32_t)d[6] >> 16;
        // the chains start from - sum(I * Ix), - sum(I * Iy) (see patch_slot); three-operand form for the
        // first link so that the constant is not copied into the accumulator first
        pb1 = m == 0 ? dot2_v(vp, IxP[m], nIIx) : dot2(vp, IxP[m], pb1);
        pb2 = m == 0 ? dot2_v(vp, IyP[m], nIIy) : dot2(vp, IyP[m], pb2);
    }
}


// One cv::calcOpticalFlowPyrLK call for the wave's four points.  Control values (prevPt, outPt,
// status, live) are per lane = per slot lane >> 4.
__device__ __forceinline__ void lk_call4(const PyrGeom &g, const uint8_t *slotI, const uint8_t *slotJ, float2 prevPt,
                                         float2 &outPt, int &status, bool live, uint32_t *lds, const uint32_t *lds_wg,
                                         int wave_off, int lane)
{
    LK_STAMP_DECL;
    LK_CALL_BEGIN;
    PixLane pl;
    pl.row = min(lane / 3, kWin - 1); pl.seg = lane - (lane / 3) * 3; pl.onmask = lane < 63 ? ~0u : 0u;
    pl.qoff = (uint32_t)((pl.seg * 7 * kQColDw + pl.row) * 4);
    asm volatile("" : "+v"(pl.onmask));                       // keep it a mask (one v_and per weight word), not a select
    const float half = 10.f;                     // (winSize - 1) * 0.5
    const float FLT_SCALE = 1.f / (1 << 20);

    uint32_t IxP[kSlots][4], IyP[kSlots][4];
    int nIIx[kSlots], nIIy[kSlots];
    // lane part of the J sample offset, the wave's LDS region included (bytes from the workgroup array)
    const int lane_off = (pl.seg * 7 * kJColDw + pl.row) * 4 + wave_off;
    int q_pr[3], q_dc4[3], jq_dst[3];   // staging item lane + 64 t = row pair * 7 + dword column
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int i = lane + 64 * t;
        q_pr[t] = i / 7; q_dc4[t] = 4 * (i - q_pr[t] * 7); jq_dst[t] = q_dc4[t] * kJColDw + q_pr[t];
    }
    const uint32_t lds_base = (uint32_t)(size_t)(lds_cu32 *)lds;   // byte address of the wave's LDS region
    int vround = 1 << (W_BITS - 5 - 1 + 7);                   // rounding of the J samples, scaled like the column words
    asm volatile("" : "+v"(vround));                          // keep it in a VGPR (see mismatch_slot)
    status = 1;
    float nx = 0.f, ny = 0.f;                    // nextPts[i]
    // The I tiles of a level depend on prevPt only: they are requested one level ahead (the top level's
    // before the loop), so their latency is covered by the previous level's iterations.
    uint32_t rI[kSlots][3][2];
    auto request_I = [&](int level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const float lscale = 1.f / (float)(1 << level);
        const int ipx = cv_floor(prevPt.x * lscale - half), ipy = cv_floor(prevPt.y * lscale - half);
        const unsigned long long m = __ballot(live && !window_oob(ipx, ipy, w, h));
        const int x0 = (ipx - 1) & ~3;
        uint32_t src[3];
#pragma unroll
        for (int t = 0; t < 3; t++) src[t] = (uint32_t)(q_pr[t] * pitch + q_dc4[t]);
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m >> (16 * s)) & 1ull)) continue;
            const int x0s = __builtin_amdgcn_readlane(x0, 16 * s), ipys = __builtin_amdgcn_readlane(ipy, 16 * s);
            // 32-bit offsets from the slot's (wave-uniform) base: scalar part per slot, lane part per level
            tile_loads(rI[s], slotI, slotI + pitch, (uint32_t)(g.origin[level] + (ipys - 1) * pitch + x0s), src, lane, kQPairs * 7);
        }
    };
    request_I(g.nlevels - 1);
    for (int level = g.nlevels - 1; level >= 0; --level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        LK_AT(0);
        // ---- control: window position and weights of every slot
        const float lscale = 1.f / (float)(1 << level);
        float px = prevPt.x * lscale, py = prevPt.y * lscale;
        if (level == g.nlevels - 1) { nx = px; ny = py; }
        else { nx = nx * 2.f; ny = ny * 2.f; }
        px -= half; py -= half;
        const int ipx = cv_floor(px), ipy = cv_floor(py);
        const bool oob = window_oob(ipx, ipy, w, h);
        if (live && oob && level == 0) status = 0;
        bool lvl_on = live && !oob;
        const PackedWeights wt = bilinear_weights(px - (float)ipx, py - (float)ipy);
        const uint32_t WIa = wt.Wa, WIb = wt.Wb;
        const int x0 = (ipx - 1) & ~3;
        const int offI = (ipx - 1) - x0;

        // ---- position of the J window at the first iteration (nextPt is known): its tile is requested
        //      together with the I tiles, so the patch arithmetic below covers the latency of both
        float qx = nx - half, qy = ny - half;       // nextPt - halfWin
        int tx0 = -(1 << 20), ty0 = 0;              // no J tile staged
        {
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (lvl_on && !window_oob(inx, iny, w, h)) { tx0 = inx - kJMargin; ty0 = iny - kJMargin; }
        }
        const unsigned long long m_on = __ballot(lvl_on), m_j = __ballot(tx0 != -(1 << 20));
        uint32_t q_src[3];                       // lane part of the tile source offsets at this level
#pragma unroll
        for (int t = 0; t < 3; t++) q_src[t] = (uint32_t)(q_pr[t] * pitch + q_dc4[t]);
        uint32_t rJ[kSlots][3][2];
        // ---- I tiles (as row-pair column words; all four fit the wave's LDS region, which the J tiles
        //      take over afterwards), patches + A sums
        int pA[kSlots][3];
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_on >> (16 * s)) & 1ull)) continue;
            uint32_t *qt = lds + s * kQTileDw;
#pragma unroll
            for (int t = 0; t < 3; t++) {
                if (lane + 64 * t < kQPairs * 7) {
                    const uint32_t top = rI[s][t][0], bot = rI[s][t][1];
                    uint32_t *d = qt + jq_dst[t];
#pragma unroll
                    for (int c = 0; c < 4; c++) d[c * kQColDw] = perm_b32(bot, top, 0x0c040c00u + 0x00010001u * c);
                }
            }
        }
        // the J tiles are requested now (vmcnt counts in order: the waits above were for the I tiles only)
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_j >> (16 * s)) & 1ull)) continue;
            const int tx0s = __builtin_amdgcn_readlane(tx0, 16 * s), ty0s = __builtin_amdgcn_readlane(ty0, 16 * s);
            tile_loads(rJ[s], slotJ, slotJ + pitch, (uint32_t)(g.origin[level] + ty0s * pitch + tx0s), q_src, lane, kJPairs * 7);
        }
        wave_lds_fence();
        LK_AT(1);                                // 0 -> 1: level control, I staging, J requests
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            pA[s][0] = pA[s][1] = pA[s][2] = 0;
            if (!((m_on >> (16 * s)) & 1ull)) continue;
            const uint32_t qaddr = lds_base + (uint32_t)((s * kQTileDw + __builtin_amdgcn_readlane(offI, 16 * s) * kQColDw) * 4);
            const uint32_t W01s = __builtin_amdgcn_readlane(WIa, 16 * s), W23s = __builtin_amdgcn_readlane(WIb, 16 * s);
            const int ipxs = __builtin_amdgcn_readlane(ipx, 16 * s), ipys = __builtin_amdgcn_readlane(ipy, 16 * s);
            if (__builtin_expect(ipxs < 0 || ipxs + kWin >= w || ipys < 0 || ipys + kWin >= h, 0))
                patch_slot<true>(qaddr, pl, W01s, W23s, ipxs, ipys, w, h, IxP[s], IyP[s], nIIx[s], nIIy[s],
                                 pA[s][0], pA[s][1], pA[s][2]);
            else
                patch_slot<false>(qaddr, pl, W01s, W23s, ipxs, ipys, w, h, IxP[s], IyP[s], nIIx[s], nIIy[s],
                                  pA[s][0], pA[s][1], pA[s][2]);
        }
        LK_AT(2);                                // 1 -> 2: patch_slot x 4
        wave_lds_fence();                        // the J tiles reuse the I tiles' LDS
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_j >> (16 * s)) & 1ull)) continue;
            tile_store_j(lds + s * kJTileDw, rJ[s], jq_dst, lane);
        }
        wave_lds_fence();
        float A11, A12, A22, D;
        {
            int v1[8], v2[8];
#pragma unroll
            for (int s = 0; s < kSlots; s++) {
                v1[2 * s] = pA[s][0]; v1[2 * s + 1] = pA[s][1];
                v2[2 * s] = pA[s][2]; v2[2 * s + 1] = 0;
            }
            const int r1 = reduce_scatter8_rows(v1, lane), r2 = reduce_scatter8_rows(v2, lane);
            A11 = wide_to_f32(quad_bcast<2>(r1), quad_bcast<0>(r1)) * FLT_SCALE;
            A12 = wide_to_f32(quad_bcast<3>(r1), quad_bcast<1>(r1)) * FLT_SCALE;
            A22 = wide_to_f32(quad_bcast<2>(r2), quad_bcast<0>(r2)) * FLT_SCALE;
        }
        D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * kWin * kWin);
        const bool degenerate = minEig < 0.001f || D < 1.1920929e-07f;
        if (lvl_on && degenerate && level == 0) status = 0;
        lvl_on = lvl_on && !degenerate;
        D = 1.f / D;

        // ---- iterations (all slots in lockstep; a slot drops out when it converges or leaves)
        if (level > 0) request_I(level - 1);
        LK_AT(3);                                // 2 -> 3: J stores, A reduction, 2x2 set-up, next level's I requests
        float pdx = 0.f, pdy = 0.f;
        bool it_on = lvl_on;
        for (int j = 0; j < kLkMaxIter; j++) {
            if (!__any(it_on)) break;
            LK_AT(4);
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (it_on && window_oob(inx, iny, w, h)) {
                if (level == 0) status = 0;
                it_on = false;
            }
            const PackedWeights wj = bilinear_weights(qx - (float)inx, qy - (float)iny);
            const uint32_t Wa = wj.Wa, Wb = wj.Wb;
            int cx = inx - tx0, cy = iny - ty0;
            const bool restage = it_on && ((unsigned)cx > (unsigned)(2 * kJMargin) || (unsigned)cy > (unsigned)(2 * kJMargin));
            if (restage) { tx0 = inx - kJMargin; ty0 = iny - kJMargin; cx = kJMargin; cy = kJMargin; }
            // slot part of the J sample offset (bytes), tile position of the slot included
            const int joff = ((int)__umul24((unsigned)cx, kJColDw) + cy + (lane >> 4) * kJTileDw) * 4;
            const unsigned long long m_it = __ballot(it_on), m_rs = __ballot(restage);
            if (__builtin_expect(m_rs != 0, 0)) {     // a window drifted out of its tile
#pragma unroll
                for (int s = 0; s < kSlots; s++) {
                    if (!((m_rs >> (16 * s)) & 1ull)) continue;
                    const int tx0s = __builtin_amdgcn_readlane(tx0, 16 * s), ty0s = __builtin_amdgcn_readlane(ty0, 16 * s);
                    uint32_t r[3][2];
                    tile_loads(r, slotJ, slotJ + pitch, (uint32_t)(g.origin[level] + ty0s * pitch + tx0s), q_src, lane, kJPairs * 7);
                    tile_store_j(lds + s * kJTileDw, r, jq_dst, lane);
                }
                wave_lds_fence();
            }
            LK_AT(5);                            // 4 -> 5: iteration control: floor, weights, restage test
            int pb[kSlots][2];
#pragma unroll
            for (int s = 0; s < kSlots; s++) {
                pb[s][0] = pb[s][1] = 0;
                if (!((m_it >> (16 * s)) & 1ull)) continue;
                const int joffs = __builtin_amdgcn_readlane(joff, 16 * s);
                const uint32_t Was = __builtin_amdgcn_readlane(Wa, 16 * s), Wbs = __builtin_amdgcn_readlane(Wb, 16 * s);
                // byte address = workgroup array + slot part (SGPR) + lane part: one add, no re-alignment of an index
                lds_cu32 *pj = (lds_cu32 *)(size_t)((uint32_t)(size_t)(lds_cu32 *)lds_wg + (uint32_t)(joffs + lane_off));
                uint32_t C[8];
#pragma unroll
                for (int k = 0; k < 8; k++) C[k] = pj[k * kJColDw];
                mismatch_slot(C, Was, Wbs, IxP[s], IyP[s], nIIx[s], nIIy[s], vround, pb[s][0], pb[s][1]);
            }
            LK_AT(6);                            // 5 -> 6: slot pixel work: J reads, bilinear + mismatch dot products
            float b1f, b2f;
            {
                int r;
#ifdef SVO_LK_SINGLE_SLOT
                // EXPERIMENT (round 5, off by default: measured SLOWER, 13.24 against 12.86 ms per 256 pairs, same box, two runs
                // each, bit-identical results): one slot left (a third of the wave-iterations: its partners have converged) --
                // its two sums alone through reduce_pair_all (13 cross-lane operations with two row swaps instead of 22 with
                // six).  The instruction arithmetic said -2.5 %; the extra scalar branch splits the iteration's one basic block
                // and costs the scheduler more overlap than the shorter reduction returns.  DESIGN.md section 6.
                if (m_it != 0 && (m_it & (m_it - 1)) == 0) {
                    // (the idle slots' partial sums are zeros: the live slot's values are the sums over the slots)
                    r = reduce_pair_all((pb[0][0] + pb[1][0]) + (pb[2][0] + pb[3][0]), (pb[0][1] + pb[1][1]) + (pb[2][1] + pb[3][1]), lane);
                } else
#endif
                {
                int v[8];
#pragma unroll
                for (int s = 0; s < kSlots; s++) { v[2 * s] = pb[s][0]; v[2 * s + 1] = pb[s][1]; }
                r = reduce_scatter8_rows(v, lane);                 // row s: {b1.lo, b2.lo, b1.hi, b2.hi} of slot s in every quad
                }
                b1f = wide_to_f32(quad_bcast<2>(r), quad_bcast<0>(r)) * FLT_SCALE;
                b2f = wide_to_f32(quad_bcast<3>(r), quad_bcast<1>(r)) * FLT_SCALE;
            }
            const float dlx = (A12 * b2f - A22 * b1f) * D;
            const float dly = (A12 * b1f - A11 * b2f) * D;
            // "delta.ddot(delta) <= epsilon" is a double comparison upstream; float decides it unless
            // the sum lands within 1e-4 relative of epsilon (float error here < 2e-7 relative)
            const float dd = dlx * dlx + dly * dly;
            bool conv = dd <= 0.9999e-4f;
            if (__builtin_expect(__any(it_on && !conv && dd < 1.0001e-4f), 0)) {
                asm volatile("" ::: "memory");               // a real branch: if-converted, the f64 path ran every iteration
                conv = (double)dlx * (double)dlx + (double)dly * (double)dly <= 0.01 * 0.01;
            }
            if (it_on) {
                qx += dlx; qy += dly;
                nx = qx + half; ny = qy + half;
                if (conv) it_on = false;
                // "std::abs(delta.x + prevDelta.x) < 0.01" compares a float with the double 0.01; the
                // largest float below 0.01 is 0.01f itself, so "<= 0.01f" in float is the same predicate
                else if (j > 0 && fabsf(dlx + pdx) <= 0.01f && fabsf(dly + pdy) <= 0.01f) {
                    nx -= dlx * 0.5f; ny -= dly * 0.5f;
                    it_on = false;
                }
                pdx = dlx; pdy = dly;
            }
            LK_AT(7);                            // 6 -> 7: reduce-scatter, solve, convergence tests
        }
        if (live && status && level == 0) {
            // err is requested by the reference: the final window must still be inside (A.4 step 7)
            int fx = cv_floor(nx - half), fy = cv_floor(ny - half);
            if (window_oob(fx, fy, w, h)) status = 0;
        }
    }
    outPt = make_float2(nx, ny);
    LK_CALL_END(lane);
}

// Grid: ONE dimension, a.gx workgroups per batch item; the workgroups of an item walk its points in
// strides of a.gx * 16 (four waves x four slots), so the launch is sized from the batch, not from
// the keypoint CAPACITY (cv::FAST is uncapped and the capacity is generous: a grid of capacity / 16
// workgroups per item was mostly empty waves).  XCD-aware mapping: consecutive workgroup ids go
// round-robin to the 8 XCDs, each with its own 4 MB L2, so item = (id / 8 / gx) * 8 + id % 8 keeps
// every XCD on its own items -- an XCD then has about one item's four pyramids (3.3 MB) in flight
// instead of slices of all items that are in flight anywhere on the chip (L2 hit rate 63 % -> see
// DESIGN.md).  No workgroup barrier anywhere: each wave loops on its own.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void lk_kernel(LkArgs a)
{
    __shared__ uint32_t lds[4 * kLdsDwPerWave];
    // items in whole groups of 8 are dealt one per XCD; the last (batch % 8) items -- the single pair of
    // the online path among them -- are spread over all XCDs in the plain order
    const int n_aware = (a.batch & ~7) * a.gx;
    int b, bx;
    if ((int)blockIdx.x < n_aware) {
        const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
        b = (slot_id / a.gx) * 8 + xcd; bx = slot_id % a.gx;
    } else {
        const int r = blockIdx.x - n_aware;
        b = (a.batch & ~7) + r / a.gx; bx = r % a.gx;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = lane >> 4;
    int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    n = min(n, a.cap);
    uint32_t *my = lds + wave * kLdsDwPerWave;
    // slots per wave: four when the launch fills the chip (the control work of an iteration is shared by
    // four points); a launch of a few items only (the online path: one pair) is latency-bound -- there the
    // points are spread over as many waves as the grid has, down to one point per wave
    int spw = kSlots;
    if (a.spread) spw = min(kSlots, max(1, (n + a.gx * 4 - 1) / (a.gx * 4)));
    for (int first = (bx * 4 + wave) * spw; first < n; first += a.gx * 4 * spw) {
        const int idx = first + slot;
        const bool valid = slot < spw && idx < n;
        const bool writer = valid && lane == 16 * slot;         // one lane per slot stores results
        const int64_t po = (int64_t)b * a.pts_stride + (valid ? idx : first);
        const float2 p0 = a.pts_in[po];
        float2 cur = p0, nxt;
        bool outside = p0.x < 0 || p0.y < 0, bad = false, noepi = false;
        bool live = valid;
        float prev_y = p0.y;
#pragma nounroll
        for (int c = 0; c < a.ncalls; c++) {
            const uint8_t *sI = a.prev[c] + (int64_t)b * a.slot_stride;
            const uint8_t *sJ = a.next[c] + (int64_t)b * a.slot_stride;
            int st;
            lk_call4(a.g, sI, sJ, cur, nxt, st, live, my, lds, wave * (kLdsDwPerWave * 4), lane);
            if (writer && live) {
                a.pts_out[c][po] = nxt;
                a.status[c][po] = (uint8_t)st;
            }
            // Tracking::deleteBadmatchFeatures terms (p0 = t1_left, p1 = t1_right, p2 = t2_right,
            // p3 = t2_left, p0_return = LK#4 output; call-site mapping src/tracking.cpp:619-620)
            if (live) {
                outside = outside || nxt.x < 0 || nxt.y < 0;
                bad = bad || st == 0;
                if (c == 0 || c == 2) noepi = noepi || (double)fabsf(prev_y - nxt.y) > a.match_err;   // |y0-y1|, |y2-y3|
                prev_y = nxt.y;
                cur = nxt;
            }
            // a rejected point can never be kept: the remaining calls of the circular chain only feed
            // the keep predicate (their pts_out/status entries are scratch in the fused mode)
            if (a.ncalls == 4 && (outside || bad || noepi)) live = false;
            if (!__any(live)) break;
        }
        if (a.ncalls == 4 && writer) a.keep[po] = !(outside || bad || noepi);
        wave_lds_fence();                                      // the next chunk restages this wave's tiles
    }
}

// Stable compaction (deleteBadmatchFeatures erases in place, preserving order): one workgroup of
// 1024 threads per batch item, ballot/popcount ranks inside waves, LDS scan across waves.
__global__ __launch_bounds__(1024) void compact_kernel(CompactArgs a)
{
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    n = min(n, a.cap);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t o = (int64_t)b * a.pts_stride;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        int i = start + threadIdx.x;
        bool k = i < n && a.keep[o + i] != 0;
        unsigned long long m = __ballot(k);
        int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int q = 0; q < 16; q++) { int t = wave_tot[q]; if (q < wv) pre += t; tot += t; }
        int dst = base_s + pre + rank;
        if (k) {
#pragma unroll
            for (int c = 0; c < 4; c++) a.out[c][o + dst] = a.in[c][o + i];
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.m_out[b] = base_s;
}

#ifdef SVO_LK_STAMP
extern "C" int svo_debug_lk_stamps(unsigned long long out[2], int reset)
{
    static unsigned long long h[2 * 1024];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lk_stamp), sizeof(h)) != hipSuccess) return -1;
    out[0] = out[1] = 0;
    for (int i = 0; i < 1024; i++) { out[0] += h[2 * i]; out[1] += h[2 * i + 1]; }
    if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_lk_stamp), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
#endif

void launch_lk(const LkArgs &a0, int batch, int max_pts, hipStream_t st)
{
    if (max_pts <= 0 || batch <= 0) return;
    if (a0.accum != 0) { launch_lk_sse2(a0, batch, max_pts, st); return; }      // SVO_LK_ACCUM_SSE2 / _SIMD128: the float-order kernel
    // up to 192 workgroups per item (3072 points per pass: a KITTI frame's ~2.5 k corners in one pass,
    // a few workgroups leave at once; denser frames loop), never more than capacity / 16
    const int chunks = (max_pts + 4 * kSlots - 1) / (4 * kSlots);
    LkArgs a = a0;
    a.gx = chunks < 192 ? chunks : 192;
    a.spread = 0;
    if (batch < 4) {                             // fewer than 768 workgroups: 3 per CU are resident at once
        static const int room_all = getenv("SVO_LK_SPREAD_ROOM") ? atoi(getenv("SVO_LK_SPREAD_ROOM")) : 768;     // test hook (A/B runs)
        const int wide = (max_pts + 3) / 4, room = room_all / batch;
        a.gx = wide < room ? wide : room;
        a.spread = 1;
    }
    a.batch = batch;
    dim3 grid(batch * a.gx, 1, 1), blk(256, 1, 1);
    hipLaunchKernelGGL(lk_kernel, grid, blk, 0, st, a);
}

void launch_compact(const CompactArgs &a, int batch, hipStream_t st)
{
    if (batch <= 0) return;
    hipLaunchKernelGGL(compact_kernel, dim3(batch), dim3(1024), 0, st, a);
}

}  // namespace svo
