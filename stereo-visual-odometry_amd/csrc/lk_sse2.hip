// lk_sse2.hip -- the pyramidal LK tracker of lk.hip with the five sums A11, A12, A22, b1, b2 accumulated in
// FLOAT, in the order of upstream's x86 build (svo_config.lk_accum = SVO_LK_ACCUM_SSE2).
//
// Why: the reference's four cv::calcOpticalFlowPyrLK calls (src/tracking.cpp:593-618) run an OpenCV 3 whose
// accumulators are float (acctype = itemtype = float outside the Tegra build) and whose SIMD block fixes the
// order of the additions: oracle/lk.c restates that order (orc_lk_set_accum(2)), and lk.hip's exact integer sums
// (canonical choice C0) differ from it in the last bits of 12 % of the track coordinates --
// enough to move RANSAC's inlier set on one pair in seven.  This kernel reproduces the x86 order bit for bit:
//   A (per level):   four float lanes q = x & 3 over x = 0..19, rows in order:  qA[q] += fx * fy   (105 terms a lane)
//                    + a scalar tail x = 20:  iA += (float)(ix * iy)  (21 terms);   A = tail + (((q0 + q1) + q2) + q3)
//   b (per iteration): _mm_madd_epi16 pairs (k, k + 4) of every group of eight pixels, x = 0..15:
//                    qb0 = [bx(0,4) by(0,4) bx(1,5) by(1,5)], qb1 = [bx(2,6) by(2,6) bx(3,7) by(3,7)], each lane
//                    += (float)(int32 pair sum) over rows and the two groups (42 terms a lane);  scalar tail x = 16..20:
//                    ib += (float)(diff * I_x) in raster order (105 terms);  bb = qb0 + qb1;  b1 = tail1 + (bb0 + bb2),
//                    b2 = tail2 + (bb1 + bb3).
// A float sum in a fixed order is a SERIAL chain: it cannot be reduced over the lanes.  What can be shared is the
// instruction stream: the wave still tracks four points ("slots", lk.hip), the 63 pixel lanes of a slot produce the
// TERMS in parallel (exact int32 products / pair sums, converted once), write them to LDS in chain order, and then
// every chain of all four slots runs in its own lane through ONE loop of 105 dependent v_add_f32.
//
// Round 5 layout (the round-4 kernel was bound by the LDS pipeline: 60 % busy, 40 % of it bank conflicts, and by a
// dependent add chain that one-and-three-quarter waves per SIMD cannot hide):
//   * pixel lanes in CHAIN ORDER: lanes 0..41 = (window row lane >> 1, SSE group lane & 1: columns 8 g .. 8 g + 7),
//     lanes 42..62 = the tail lanes (row lane - 42, columns 16..20).  A group lane's term t = lane of every lane
//     chain: its stores are 42 consecutive dwords, and with a column stride of 26 words the tile reads of lanes
//     0..31 fall on 32 different banks (16 g + row); the other half is 2-way on ten banks (three accesses to a bank
//     are forced by the 8 + 8 + 5 split);
//   * chain lanes: row s of 16 lanes = slot s; position 0 / 8 run the x / y tail, 1..4 and 9..12 the eight lane
//     chains.  Phase 1 (terms 0..43): every chain lane reads its own terms (11 ds_read_b128).  Phase 2 (terms
//     44..104: tails only): the EIGHT lanes of a half row each fetch four terms of their tail per read and the
//     tail lane adds them through the DPP network (v_add_f32 row_shl:f) -- two reads instead of sixteen: 13
//     wide reads per wave-iteration instead of 27, conflict-free in the hardware's four read groups (LDS plan below);
//   * A: five lanes per slot (four SSE lanes + the tail), each carrying A11, A12, A22 at once: two SDWA
//     conversions, one v_pk_fma_f32 and one v_fma_f32 per term (a product of two patch values is < 2^24, exact
//     in float, so the fused form rounds exactly as _mm_mul_ps + _mm_add_ps do), three independent chains a lane;
//   * tiles: 26 columns x 26 words (margin 2 around the J window): 20 096 B of LDS per wave, EIGHT waves per CU --
//     two per SIMD, so a dependent chain of one wave issues beside the other's.
// Everything else -- tiles as row-pair column words, packed Scharr on the fly, weights, control flow, the circular
// chain of four calls, the keep predicate -- is lk.hip's, and so are the fixed-point pixel values (integers: exact).
#include "lk_common.h"

namespace svo {

// ---- LDS plan of one wave (dwords) ----------------------------------------------------------------------------
// tiles: four slots x 26 columns x 26 words, column-major (I: 23 row pairs x 24 columns from image column ipx - 1 --
// fetched unaligned, so no alignment slack --, J: 25 pairs x 26 columns = window + margin 2 on every side);
// a staged source dword carries four columns, so the seventh dword of a row spills two columns past the tile: into
// the next slot's first two columns (which that slot's own stores, issued later, overwrite) or, behind slot 3, into
// the first 52 words of the chain staging, which are dead while tiles are staged at a level's start (the A words start
// behind them).  A re-stage of one slot during the iterations masks those two stores instead (tile_store_j2<true>).
constexpr int kCS2 = 26, kJMargin2 = 2;
constexpr int kQPairs2 = 23, kJPairs2 = kWin + 2 * kJMargin2;                       // 23, 25 (<= kCS2)
constexpr int kTileDw2 = 26 * kCS2, kTileSpill = 2 * kCS2;                         // 676, 52
constexpr int kTilesDw = kSlots * kTileDw2;                                        // 2704
// Chain staging of one slot, 576 words.  b (per iteration): the eight lane chains c = 2 k + (x|y) (42 terms + 2 the
// chain reads but does not use) and the two tails (105 terms, 108 read) at the offsets below.  A ds_read_b128 is served
// in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- on 64 banks
// (MI355X_MICROARCH.md, LDS): with these offsets, a slot stride of 9 x 64 words and the chain lanes placed as in
// make_lane, the eleven reads of phase 1 and the two of phase 2 are conflict-free in every group (found by search over
// the orders: tools/model/lds_chain_layout.py).  A (per level, same area, behind the 52 words a tile may spill into):
// patch words Ix | Iy << 16 in the order of the A chains (below).
constexpr int kStageDw = 576;
constexpr int kTailY = 0, kTailX = 416;
__device__ constexpr int kChainOff[8] = {108, 196, 372, 284, 328, 240, 524, 152};
// A staging: SSE lane q at kAq0 + q * stride, the scalar tail behind the four lanes.  SSE2 order (a.accum = 1): four lanes
// over x = 0..19 (105 words a lane, stride 108), tail x = 20 (21 words); SIMD128 order (a.accum = 2: the universal-intrinsic
// block works in groups of EIGHT pixels): four lanes over x = 0..15 (84 words, stride 84), tail x = 16..20 (105 words).
constexpr int kAq0 = kTileSpill;
__host__ __device__ constexpr int a_stride(bool simd128) { return simd128 ? 84 : 108; }
__host__ __device__ constexpr int a_tail_base(bool simd128) { return kAq0 + 4 * a_stride(simd128); }                   // 484 | 388
constexpr int kLdsDwSse2 = kTilesDw + kSlots * kStageDw + 16;                      // 5024 dwords = 20 096 B (16: the A tail lane reads on past its words)
constexpr int kDumpB = kChainOff[0] + 43, kDumpA = kStageDw - 1;                    // entries no chain uses: padding of lane chain 0 / the last word of the area
// LEGACY b order (a.accum = 3, SVO_LK_ACCUM_SSE2_LEGACY = oracle mode 3: upstream's older CV_SSE2 block multiplies pixel by pixel --
// _mm_mullo / _mm_mulhi_epi16 of (It_k It_k) x (Ix_k Iy_k) -- and adds the converted products one by one: pixels 0, 1, then 4, 5 of a
// group of eight to qb0, 2, 3, then 6, 7 to qb1).  The eight lane chains are the same lanes as above, but a chain takes TWO terms per
// group -- pixel k, then pixel k + 4 -- instead of their exact int32 sum: 84 terms (88 read), term t = 4 row + 2 group + (0 | 1).  A
// slot's staging is 960 words, six waves per CU.  Round 6: the block order below and the slot stride come from the same search as
// the madd orders' (tools/model/lds_chain_layout.py legacy): the 22 chain reads and the feeder read are conflict-free in the
// hardware's four read groups -- in the plain order of round 5 (tails at 0 / 108, chain c at 216 + 88 c, stride 928) every one of
// them took 8 LDS cycles instead of 4 (profiles/r06_lk_sse2_legacy_r05src_pmc.json: 4.96 G conflict cycles of 11.17 G).
constexpr int kStageDwL = 960, kTailYL = 180, kTailXL = 468;
__device__ constexpr int kChainOffL[8] = {88, 584, 0, 764, 676, 288, 376, 860};
constexpr int kLdsDwLegacy = kTilesDw + kSlots * kStageDwL + 16;                   // 6560 dwords = 26 240 B
constexpr int kDumpBL = kChainOffL[0] + 87, kDumpAL = kStageDwL - 1;                // padding of lane chain 0 / the last word of the area
static_assert(kChainOffL[7] + 88 <= kStageDwL - 1 && a_tail_base(false) + 24 < kStageDwL && kLdsDwLegacy * 4 <= 26 * 1024 + 512, "legacy staging");
template <bool LEGACY> __host__ __device__ constexpr int stage_dw() { return LEGACY ? kStageDwL : kStageDw; }
static_assert(kStageDw % 64 == 0 && kTailX % 4 == 0 && kTailY % 4 == 0 && kTilesDw % 4 == 0 && kAq0 % 4 == 0 && a_stride(false) % 4 == 0 && a_stride(true) % 4 == 0,
              "chain reads are 16-byte loads; the conflict-free order assumes a slot stride of whole bank rounds");
static_assert(a_tail_base(false) + 24 < kStageDw && a_tail_base(true) + 108 < kStageDw && kJPairs2 <= kCS2 && kQPairs2 <= kCS2, "staging / tile geometry");
static_assert(kLdsDwSse2 * 4 <= 20480, "eight single-wave workgroups per CU");

typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(3))) lds_cf32x4;
typedef const u32x4 __attribute__((address_space(3))) lds_cu32x4;

// per-lane constants: the pixel role and the chain role (position lane & 15 of row `slot`)
struct Sse2Lane {
    int row, x0;            // window row, first window column (0, 8: the SSE groups; 16: the tail)
    uint32_t onmask;        // all ones in the 63 pixel lanes
    uint32_t pmask;         // mask of the patch pairs: a tail lane owns pixels 0..4 only (high halves = pixels 4..7 leave the pairs)
    uint32_t t4mask;        // tail lanes: the high half of pair 0 (pixel 4, x = 20) as a patch pair of its own; group lanes: 0
    uint32_t qoff;          // byte offset of the lane's first tile word inside a slot tile (I and J tiles share the layout)
    uint32_t wb[16];        // b term j of slot 0 goes to LDS byte address wb[j] (ten terms; sixteen in the legacy order)
    uint32_t wa[8];         // patch word i (pixel x0 + i) of slot 0 goes to LDS byte address wa[i]
    uint32_t cb, cbA, cbB;  // b chain reads: own terms 0..43; the half row's tail terms 44 + 4 f.., 76 + 4 f.. (f = lane & 7)
    uint32_t ca;            // A chain read address
    bool b_tail;            // position 0 / 8 runs a b tail
    int a_sel;              // which running A sums this lane hands over: after 21, 84 or all 105 terms (0 / 1 / 2)
};

__device__ __forceinline__ void lds_store(uint32_t byte_addr, uint32_t v) { *(lds_u32 *)(size_t)byte_addr = v; }

template <bool LEGACY>
__device__ __forceinline__ Sse2Lane make_lane(int lane, uint32_t lds_base, bool simd128)
{
    Sse2Lane L;
    const bool group = lane < 42, on = lane < 63, tail = on && !group;
    L.row = group ? lane >> 1 : (tail ? lane - 42 : kWin - 1);
    L.x0 = group ? 8 * (lane & 1) : 16;
    L.onmask = on ? ~0u : 0u;
    L.pmask = group ? ~0u : 0x0000FFFFu;
    L.t4mask = tail ? 0xFFFF0000u : 0u;
    L.qoff = (uint32_t)((L.x0 * kCS2 + L.row) * 4);
    const uint32_t stage0 = lds_base + kTilesDw * 4;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        int e;
        if (!LEGACY) {
            e = kDumpB;
            if (group && j < 8) e = kChainOff[j] + lane;                                // lane chain j, term t = lane
            if (tail && j < 10) e = ((j & 1) ? kTailY : kTailX) + 5 * L.row + (j >> 1); // tail x | y, term 5 row + i
        } else {
            // term j = 2 m + xy: pixel m of the lane's eight (m < 4), j = 8 + 2 m + xy: pixel m + 4
            const int c = j & 7, second = j >> 3;
            e = kDumpBL;
            if (group) e = kChainOffL[c] + 4 * L.row + 2 * (lane & 1) + second;
            if (tail && j < 8) e = ((j & 1) ? kTailYL : kTailXL) + 5 * L.row + (j >> 1);              // x = 16..19
            if (tail && (j == 8 || j == 9)) e = ((j & 1) ? kTailYL : kTailXL) + 5 * L.row + 4;         // x = 20
        }
        L.wb[j] = stage0 + (uint32_t)e * 4;
    }
    // patch words.  SSE2 order: pixel x -> W[x & 3][5 row + (x >> 2)] for x < 20, tail word `row` for x = 20.
    // SIMD128 order: W[x & 3][4 row + (x >> 2)] for x < 16, tail word 5 row + (x - 16) for x = 16..20.  x > 20: nowhere.
    const int aq = a_stride(simd128), at = a_tail_base(simd128), per_row = simd128 ? 4 : 5;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int x = L.x0 + i;
        int e = LEGACY ? kDumpAL : kDumpA;
        if (group || (tail && !simd128 && i < 4)) e = kAq0 + (x & 3) * aq + per_row * L.row + (x >> 2);
        if (tail && !simd128 && i == 4) e = at + L.row;
        if (tail && simd128 && i < 5) e = at + 5 * L.row + i;
        L.wa[i] = stage0 + (uint32_t)e * 4;
    }
    // chain roles.  Positions 0 / 8: the x / y tail; 1..4: lane chains 0, 4, 2, 6 (the x sums); 9..12: chains 1, 5, 3, 7;
    // an idle position reads what a busy lane of ITS read group reads (one broadcast access): 5..7 follow position 4,
    // 13..15 position 12
    const int p = lane & 15, f = p & 7, h = p >> 3;
    const uint32_t stage_s = stage0 + (uint32_t)((lane >> 4) * stage_dw<LEGACY>()) * 4;
    const int tail_dw = LEGACY ? (h ? kTailYL : kTailXL) : (h ? kTailY : kTailX);
    L.b_tail = f == 0;
    const int fc = f == 0 ? 0 : (f < 4 ? f : 4);                      // 1..4: the lane chain whose terms this position reads
    const int cmap = ((fc - 1) & 1) * 4 + ((fc - 1) >> 1) * 2 + h;
    L.cb = stage_s + (uint32_t)(fc == 0 ? tail_dw : (LEGACY ? kChainOffL[fc == 0 ? 0 : cmap] : kChainOff[fc == 0 ? 0 : cmap])) * 4;
    L.cbA = stage_s + (uint32_t)(tail_dw + (LEGACY ? 88 : 44) + 4 * f) * 4;
    L.cbB = L.cbA + 32 * 4;
    // A: positions 0..3 the SSE lanes, 4 the tail; 5..11 follow position 4, 12..15 position 0
    L.a_sel = simd128 ? (p == 4 ? 2 : 1) : (p == 4 ? 0 : 2);
    L.ca = stage_s + (uint32_t)(p < 4 ? kAq0 + p * aq : (p < 12 ? at : kAq0)) * 4;
    return L;
}

// DPP inside a row of 16: the value of the lane N above (row_shl; lanes past the row's end read 0)
template <int N>
__device__ __forceinline__ float row_shl(float v)
{
    if (N == 0) return v;
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true));
}
// lane 0 of the row in every lane of the row: two masked row shifts carry quad 0 to the other three quads, a quad
// broadcast picks the lane
__device__ __forceinline__ float row_first(float v)
{
    int x = __float_as_int(v);
    x = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xF, 0x2, false);      // row_shr:4 into bank 1 (lanes 4..7)
    x = __builtin_amdgcn_update_dpp(x, x, 0x118, 0xF, 0xC, false);      // row_shr:8 into banks 2, 3 (lanes 8..15)
    return __int_as_float(quad_bcast<0>(x));
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c)       // two fused multiply-adds in one instruction
{
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// A staged source dword pair's four column words into an I tile (cf. lk.hip); the seventh dword's last two columns
// land past the tile (see the LDS plan)
__device__ __forceinline__ void tile_store_i2(uint32_t *tile, const uint32_t (&r)[3][2], const int (&q_dst)[3], int lane)
{
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (lane + 64 * t < kQPairs2 * 7) {
            const uint32_t top = r[t][0], bot = r[t][1];
            uint32_t *d = tile + q_dst[t];
#pragma unroll
            for (int c = 0; c < 4; c++) d[c * kCS2] = perm_b32(bot, top, 0x0c040c00u + 0x00010001u * c);
        }
    }
}
// ... into a J tile (samples as pixel << 7, lk_common.h).  MASKED: the re-stage of ONE slot beside live tiles keeps the
// two columns past the tile away from its neighbour.
template <bool MASKED>
__device__ __forceinline__ void tile_store_j2(uint32_t *tile, const uint32_t (&r)[3][2], const int (&q_dst)[3], const int (&q_dc4)[3], int lane)
{
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (lane + 64 * t < kJPairs2 * 7) {
            const uint32_t top = r[t][0], bot = r[t][1];
            uint32_t *d = tile + q_dst[t];
            const u16x2 one = {1, 1};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (MASKED && c >= 2 && q_dc4[t] == 24) continue;
                d[c * kCS2] = as_u32(as_u16x2(perm_b32(bot, top, 0x040c000cu + 0x01000100u * c)) >> one);
            }
        }
    }
}

// ---- phase A for one slot: the lane's 8 patch pixels from the staged I tile (cf. patch_slot in lk.hip) ------------
// Tile bytes j = 0..10 of the lane = image columns ipx - 1 + x0 + j.  Outputs: the patch packed as madd pairs
// (pixel m | pixel m + 4 << 16, m = 0..3) -- I with 5 fractional bits, Ix, Iy -- and the 8 patch words Ix | Iy << 16
// of the A chains.  Pixels right of column 20 (tail lanes, i > 4) are computed from whatever lies beside the tile and
// masked out of the pairs; their patch words go to a dump entry.  A tail lane's pixel 4 (x = 20) is a scalar-tail pixel of
// its own: it leaves pair 0 and comes back as (0 | value << 16) in Ix4 / Iy4 (zero in the group lanes).
template <bool EDGE>
__device__ __forceinline__ void patch_slot8(uint32_t tile_addr, const Sse2Lane &L, uint32_t Wau, uint32_t Wbu, int ipx,
                                            int ipy, int w, int h, uint32_t (&IvP)[4], uint32_t (&IxP)[4],
                                            uint32_t (&IyP)[4], uint32_t &Ix4, uint32_t &Iy4, uint32_t (&Aw)[8])
{
    const uint32_t Wa = Wau & L.onmask, Wb = Wbu & L.onmask;
    uint32_t Q01[11], Q12[11], Q23[11];
    {
        lds_cu32 *q0 = (lds_cu32 *)(size_t)(tile_addr + L.qoff);
#pragma unroll
        for (int j = 0; j < 11; j++) { Q01[j] = q0[j * kCS2]; Q12[j] = q0[j * kCS2 + 1]; Q23[j] = q0[j * kCS2 + 2]; }
    }
    uint32_t T0[11], T1[11];
    const u16x2 k12 = {12, 12}, k40 = {40, 40};
#pragma unroll
    for (int j = 0; j < 11; j++) {
        T0[j] = as_u32((as_u16x2(Q01[j]) + as_u16x2(Q23[j])) * k12 + as_u16x2(Q12[j]) * k40);       // 4 t0
        T1[j] = as_u32(as_u16x2(Q23[j]) - as_u16x2(Q01[j]));                                          // t1
    }
    uint32_t DX[9], DY[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        DX[c] = as_u32(as_u16x2(T0[c + 2]) - as_u16x2(T0[c]));                                        // 4 dx
        DY[c] = as_u32((as_u16x2(T1[c]) + as_u16x2(T1[c + 2])) * k12 + as_u16x2(T1[c + 1]) * k40);    // 4 dy
    }
    if (EDGE) {                                  // the derivative image's border is BORDER_CONSTANT 0
        const int gyA = ipy + L.row, gyB = gyA + 1;
        const uint32_t rows = ((gyA >= 0 && gyA < h) ? 0x0000FFFFu : 0u) | ((gyB >= 0 && gyB < h) ? 0xFFFF0000u : 0u);
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const int gx = ipx + L.x0 + c;
            const uint32_t mk = (gx >= 0 && gx < w) ? rows : 0u;
            DX[c] &= mk; DY[c] &= mk;
        }
    }
    int iv[8], ix[8], iy[8];                     // ix, iy: value << 16 | rounding residue
#pragma unroll
    for (int k = 0; k < 8; k++) {
        iv[k] = dot2(Q12[k + 2], Wb, dot2_k(Q12[k + 1], Wa, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        ix[k] = dot2(DX[k + 1], Wb, dot2_k(DX[k], Wa, 1 << (W_BITS + 1)));
        iy[k] = dot2(DY[k + 1], Wb, dot2_k(DY[k], Wa, 1 << (W_BITS + 1)));
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        IvP[m] = perm_b32((uint32_t)iv[m + 4], (uint32_t)iv[m], 0x05040100u);
        IxP[m] = perm_b32((uint32_t)ix[m + 4], (uint32_t)ix[m], 0x07060302u);      // the two high halves
        IyP[m] = perm_b32((uint32_t)iy[m + 4], (uint32_t)iy[m], 0x07060302u);
        if (m == 0) { Ix4 = IxP[0] & L.t4mask; Iy4 = IyP[0] & L.t4mask; }      // a tail lane's pixel 4 (x = 20) alone
        IxP[m] &= L.pmask; IyP[m] &= L.pmask;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) Aw[i] = perm_b32((uint32_t)iy[i], (uint32_t)ix[i], 0x07060302u);   // Ix | Iy << 16
}

// ---- one iteration's pixel work for one slot: the lane's ten b terms as floats -----------------------------------
// group lanes:  v[2 k + xy] = (float)(diff_k I_k + diff_k+4 I_k+4)  -- the int32 lanes of _mm_madd_epi16, converted as
//               _mm_cvtepi32_ps does (round to nearest even);  v[8], v[9] = 0 (stored to a dump entry)
// tail lanes:   v[2 i + xy] = (float)(diff_i I_i) for the tail pixels i = 0..3 (x = 16..19: the pairs' high halves are
//               masked out of the patch), v[8 + xy] for i = 4 (x = 20: the high half of pair 0 against Ix4 / Iy4)
// LEGACY (the older CV_SSE2 block): every product on its own -- v[2 m + xy] = (float)(diff_m I_m), v[8 + 2 m + xy] = (float)(diff_m+4
// I_m+4) (group lanes: sixteen chain terms; tail lanes: pixels 0..3 in v[0..7], x = 20 in v[8 + xy], the rest zero)
template <bool LEGACY>
__device__ __forceinline__ void mismatch_slot8(const uint32_t (&C)[9], uint32_t Wa, uint32_t Wb, const uint32_t (&IvP)[4],
                                               const uint32_t (&IxP)[4], const uint32_t (&IyP)[4], uint32_t Ix4, uint32_t Iy4,
                                               int vround, float (&v)[16])
{
    int d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = dot2(C[k + 1], Wb, dot2_sv(C[k], Wa, vround));
    uint32_t df[4];                              // diff = J - I of pixels (m, m + 4): the J samples are the high halves of d
#pragma unroll
    for (int m = 0; m < 4; m++) {
        df[m] = as_u32(as_u16x2(perm_b32((uint32_t)d[m + 4], (uint32_t)d[m], 0x07060302u)) - as_u16x2(IvP[m]));
        if (!LEGACY) {
            v[2 * m] = (float)dot2_0(df[m], IxP[m]);
            v[2 * m + 1] = (float)dot2_0(df[m], IyP[m]);
        } else {
            // the pair's two products apart: the other half of the difference word masked away (a tail lane's pixel 4 rides Ix4 / Iy4:
            // zero in the group lanes, whose pair 0 keeps its own high half)
            const uint32_t lo = df[m] & 0x0000FFFFu, hi = df[m] & 0xFFFF0000u;
            const uint32_t ixh = m == 0 ? (IxP[0] | Ix4) : IxP[m], iyh = m == 0 ? (IyP[0] | Iy4) : IyP[m];
            v[2 * m] = (float)dot2_0(lo, IxP[m]);
            v[2 * m + 1] = (float)dot2_0(lo, IyP[m]);
            v[8 + 2 * m] = (float)dot2_0(hi, ixh);
            v[8 + 2 * m + 1] = (float)dot2_0(hi, iyh);
        }
    }
    if (!LEGACY) { v[8] = (float)dot2_0(df[0], Ix4); v[9] = (float)dot2_0(df[0], Iy4); }
}

// ---- the serial sums: every chain lane adds its chain's terms in order -------------------------------------------
// b.  Phase 1: terms 0..43 of the lane's own chain (a lane chain has 42; its sum is taken before the two padding terms).
// Phase 2: the tails' terms 44..104 -- the eight lanes of a half row hold four consecutive terms each (two reads), and the
// tail lane at the head of the half row adds them in order over the DPP network; the other lanes add garbage.
template <int F>
__device__ __forceinline__ void tail_adds(float &acc, const f32x4 &q, int first_term)
{
#pragma unroll
    for (int e = 0; e < 4; e++)
        if (first_term + 4 * F + e < 105) acc += row_shl<F>(q[e]);
}
template <bool LEGACY>
__device__ __forceinline__ float chain_b(const Sse2Lane &L)
{
    lds_cf32x4 *p = (lds_cf32x4 *)(size_t)L.cb;
    if (LEGACY) {
        // 22 reads: the 84 terms of a lane chain (+ 4 of padding), the first 88 of a tail; terms 88..104 of the tails over the DPP
        // network from five feeder lanes
        const f32x4 qa = *(lds_cf32x4 *)(size_t)L.cbA;
        float acc = 0.f, acc84 = 0.f;
#pragma unroll
        for (int t = 0; t < 22; t++) {
            const f32x4 q = p[t];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (4 * t + e == 84) acc84 = acc;
                acc += q[e];
            }
        }
        tail_adds<0>(acc, qa, 88); tail_adds<1>(acc, qa, 88); tail_adds<2>(acc, qa, 88); tail_adds<3>(acc, qa, 88); tail_adds<4>(acc, qa, 88);
        return L.b_tail ? acc : acc84;
    }
    const f32x4 qa = *(lds_cf32x4 *)(size_t)L.cbA, qb = *(lds_cf32x4 *)(size_t)L.cbB;
    float acc = 0.f, acc42 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; t++) {
        const f32x4 q = p[t];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (4 * t + e == 42) acc42 = acc;
            acc += q[e];
        }
    }
    tail_adds<0>(acc, qa, 44); tail_adds<1>(acc, qa, 44); tail_adds<2>(acc, qa, 44); tail_adds<3>(acc, qa, 44);
    tail_adds<4>(acc, qa, 44); tail_adds<5>(acc, qa, 44); tail_adds<6>(acc, qa, 44); tail_adds<7>(acc, qa, 44);
    tail_adds<0>(acc, qb, 76); tail_adds<1>(acc, qb, 76); tail_adds<2>(acc, qb, 76); tail_adds<3>(acc, qb, 76);
    tail_adds<4>(acc, qb, 76); tail_adds<5>(acc, qb, 76); tail_adds<6>(acc, qb, 76); tail_adds<7>(acc, qb, 76);
    return L.b_tail ? acc : acc42;
}
#ifdef SVO_LKS_EXPERIMENTS
// TIMING EXPERIMENTS of round 6 (never in the shipped build: tools/gpu/lks_chain_bound.sh compiles them in on the GPU box and
// selects one with SVO_LKS_EXP; results in profiles/r06_lk_sse2_chain_bound.json).  Question: what would an integer / tree
// fast path for the b chains buy when the float sums are provably exact?  chain_b_tree adds the same staged terms as a TREE --
// four accumulators over phase 1, every lane of a half row its own eight tail terms, three DPP steps -- which is what ANY
// exact fast path has to do at least (its sums differ from the serial order in the last bit when a partial sum passes 2^24:
// a bound, not a product path), optionally with the sums of |terms| beside it (a guard evaluated in the chain lanes).
template <bool GUARD>
__device__ __forceinline__ float chain_b_tree(const Sse2Lane &L, int lane, float &guard)
{
    lds_cf32x4 *p = (lds_cf32x4 *)(size_t)L.cb;
    const f32x4 qa = *(lds_cf32x4 *)(size_t)L.cbA, qb = *(lds_cf32x4 *)(size_t)L.cbB;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 10; t++) {
        const f32x4 q = p[t];
#pragma unroll
        for (int e = 0; e < 4; e++) { s[e] += q[e]; if (GUARD) a[e] += __builtin_fabsf(q[e]); }
    }
    const f32x4 q10 = p[10];
    s[0] += q10[0]; s[1] += q10[1];
    if (GUARD) { a[0] += __builtin_fabsf(q10[0]); a[1] += __builtin_fabsf(q10[1]); }
    const float pad_s = q10[2] + q10[3], pad_a = __builtin_fabsf(q10[2]) + __builtin_fabsf(q10[3]);
    float S = (s[0] + s[1]) + (s[2] + s[3]), A = (a[0] + a[1]) + (a[2] + a[3]);
    const int f = lane & 7;
    const bool v1 = f < 7;
    float ts = (qa[0] + qa[1]) + (qa[2] + qa[3]) + (qb[0] + (v1 ? (qb[1] + qb[2]) + qb[3] : 0.f));
    float ta = 0.f;
    if (GUARD) ta = (__builtin_fabsf(qa[0]) + __builtin_fabsf(qa[1])) + (__builtin_fabsf(qa[2]) + __builtin_fabsf(qa[3])) +
                    (__builtin_fabsf(qb[0]) + (v1 ? (__builtin_fabsf(qb[1]) + __builtin_fabsf(qb[2])) + __builtin_fabsf(qb[3]) : 0.f));
    ts += row_shl<4>(ts); ts += row_shl<2>(ts); ts += row_shl<1>(ts);
    if (GUARD) { ta += row_shl<4>(ta); ta += row_shl<2>(ta); ta += row_shl<1>(ta); }
    guard = L.b_tail ? A + pad_a + ta : A;
    return L.b_tail ? S + pad_s + ts : S;
}
#endif
// A: positions 0..3 of a row = the SSE lanes q (105 terms), position 4 = the scalar tail (21 terms); every lane carries
// all three sums.  A term is the product of two 16-bit patch values, < 2^24: exact in float, so fma(fx, fy, acc) rounds
// once exactly where _mm_add_ps(acc, _mm_mul_ps(fx, fy)) / iA += (float)(ix * iy) round.
// terms first..last-1 of a block of three reads (twelve patch words Ix | Iy << 16)
__device__ __forceinline__ void a_terms(f32x2 &d, float &m, const u32x4 (&q)[3], int first, int last)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (4 * i + e >= first && 4 * i + e < last) {
                const f32x2 f = {(float)(short)(q[i][e] & 0xFFFFu), (float)((int)q[i][e] >> 16)};
                d = pk_fma(f, f, d);
                m = __builtin_fmaf(f.x, f.y, m);
            }
}
__device__ __forceinline__ void chain_a(const Sse2Lane &L, float &s11, float &s12, float &s22)
{
    lds_cu32x4 *p = (lds_cu32x4 *)(size_t)L.ca;
    f32x2 d = {0.f, 0.f}, d21, d84 = {0.f, 0.f};    // (A11, A22)
    float m = 0.f, m21, m84 = 0.f;                   // A12
    // Three reads (twelve terms) a block, the next block requested before this one is added.  The middle of the chain is
    // a real loop of two blocks a turn: fully unrolled, the function crossed a size at which the register allocator gave
    // up on the kernel (150 spilled registers, the next level's tile requests among them).  Every lane runs all 105 terms;
    // which running sum it hands over depends on its chain: 21 terms (the SSE2 order's tail), 84 (the SIMD128 order's
    // four lanes) or all of them.
    u32x4 q[3], n[3];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = p[i];
#pragma unroll
    for (int i = 0; i < 3; i++) n[i] = p[3 + i];
    asm volatile("" ::: "memory");
    a_terms(d, m, q, 0, 12);                    // terms 0..11
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = p[6 + i];
    asm volatile("" ::: "memory");
    a_terms(d, m, n, 0, 9);                     // 12..20
    d21 = d; m21 = m;
    a_terms(d, m, n, 9, 12);                    // 21..23
#pragma nounroll
    for (int turn = 0; turn < 3; turn++, p += 6) {     // q = terms 24 + 24 turn ..; read ahead by one block
#pragma unroll
        for (int i = 0; i < 3; i++) n[i] = p[9 + i];
        asm volatile("" ::: "memory");
        a_terms(d, m, q, 0, 12);
        if (turn == 2) { d84 = d; m84 = m; }    // terms 0..83
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = p[12 + i];
        asm volatile("" ::: "memory");
        a_terms(d, m, n, 0, 12);
    }
    a_terms(d, m, q, 0, 9);                     // 96..104
    s11 = L.a_sel == 0 ? d21.x : (L.a_sel == 1 ? d84.x : d.x);
    s12 = L.a_sel == 0 ? m21 : (L.a_sel == 1 ? m84 : m);
    s22 = L.a_sel == 0 ? d21.y : (L.a_sel == 1 ? d84.y : d.y);
}
// tail + (((q0 + q1) + q2) + q3) at position 0 of the row, handed to the whole row
__device__ __forceinline__ float combine_a(float r)
{
    const float u = ((r + row_shl<1>(r)) + row_shl<2>(r)) + row_shl<3>(r);
    return row_first(row_shl<4>(r) + u);
}

// One cv::calcOpticalFlowPyrLK call for the wave's four points (cf. lk_call4 in lk.hip; control values are per
// lane = per slot lane >> 4).
template <bool LEGACY, int EXP>
__device__ __forceinline__ void lk_call4_sse2(const PyrGeom &g, const uint8_t *slotI, const uint8_t *slotJ, float2 prevPt,
                                              float2 &outPt, int &status, bool live, uint32_t *lds, int lane,
                                              const Sse2Lane &L)
{
    const float half = 10.f;                     // (winSize - 1) * 0.5
    const float FLT_SCALE = 1.f / (1 << 20);
    uint32_t IvP[kSlots][4], IxP[kSlots][4], IyP[kSlots][4], Ix4[kSlots], Iy4[kSlots];
    int q_pr[3], q_dc4[3], q_dst[3];             // staging item lane + 64 t = row pair * 7 + dword column
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int i = lane + 64 * t;
        q_pr[t] = i / 7; q_dc4[t] = 4 * (i - q_pr[t] * 7); q_dst[t] = q_dc4[t] * kCS2 + q_pr[t];
    }
    const uint32_t lds_base = (uint32_t)(size_t)(lds_cu32 *)lds;
    int vround = 1 << (W_BITS - 5 - 1 + 7);
    asm volatile("" : "+v"(vround));
    status = 1;
    float nx = 0.f, ny = 0.f;
    uint32_t rI[kSlots][3][2];
    auto request_I = [&](int level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const float lscale = 1.f / (float)(1 << level);
        const int ipx = cv_floor(prevPt.x * lscale - half), ipy = cv_floor(prevPt.y * lscale - half);
        const unsigned long long m = __ballot(live && !window_oob(ipx, ipy, w, h));
        const int x0 = ipx - 1;                  // unaligned, like the J tiles: the patch needs exactly 24 columns from here
        uint32_t src[3];
#pragma unroll
        for (int t = 0; t < 3; t++) src[t] = (uint32_t)(q_pr[t] * pitch + q_dc4[t]);
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m >> (16 * s)) & 1ull)) continue;
            const int x0s = __builtin_amdgcn_readlane(x0, 16 * s), ipys = __builtin_amdgcn_readlane(ipy, 16 * s);
            tile_loads(rI[s], slotI, slotI + pitch, (uint32_t)(g.origin[level] + (ipys - 1) * pitch + x0s), src, lane, kQPairs2 * 7);
        }
    };
    request_I(g.nlevels - 1);
    for (int level = g.nlevels - 1; level >= 0; --level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
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
        float qx = nx - half, qy = ny - half;       // nextPt - halfWin
        int tx0 = -(1 << 20), ty0 = 0;              // no J tile staged
        {
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (lvl_on && !window_oob(inx, iny, w, h)) { tx0 = inx - kJMargin2; ty0 = iny - kJMargin2; }
        }
        const unsigned long long m_on = __ballot(lvl_on), m_j = __ballot(tx0 != -(1 << 20));
        uint32_t q_src[3];
#pragma unroll
        for (int t = 0; t < 3; t++) q_src[t] = (uint32_t)(q_pr[t] * pitch + q_dc4[t]);
        uint32_t rJ[kSlots][3][2];
        // ---- I tiles as row-pair column words (the J tiles take their place afterwards)
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_on >> (16 * s)) & 1ull)) continue;
            tile_store_i2(lds + s * kTileDw2, rI[s], q_dst, lane);
        }
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_j >> (16 * s)) & 1ull)) continue;
            const int tx0s = __builtin_amdgcn_readlane(tx0, 16 * s), ty0s = __builtin_amdgcn_readlane(ty0, 16 * s);
            tile_loads(rJ[s], slotJ, slotJ + pitch, (uint32_t)(g.origin[level] + ty0s * pitch + tx0s), q_src, lane, kJPairs2 * 7);
        }
        wave_lds_fence();
        // ---- patches; their words Ix | Iy << 16 go to the chain staging in the order of the A chains
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_on >> (16 * s)) & 1ull)) continue;
            const uint32_t qaddr = lds_base + (uint32_t)(s * kTileDw2 * 4);
            const uint32_t W01s = __builtin_amdgcn_readlane(WIa, 16 * s), W23s = __builtin_amdgcn_readlane(WIb, 16 * s);
            const int ipxs = __builtin_amdgcn_readlane(ipx, 16 * s), ipys = __builtin_amdgcn_readlane(ipy, 16 * s);
            uint32_t Aw[8];
            if (__builtin_expect(ipxs < 0 || ipxs + kWin >= w || ipys < 0 || ipys + kWin >= h, 0))
                patch_slot8<true>(qaddr, L, W01s, W23s, ipxs, ipys, w, h, IvP[s], IxP[s], IyP[s], Ix4[s], Iy4[s], Aw);
            else
                patch_slot8<false>(qaddr, L, W01s, W23s, ipxs, ipys, w, h, IvP[s], IxP[s], IyP[s], Ix4[s], Iy4[s], Aw);
            const uint32_t so = (uint32_t)(s * stage_dw<LEGACY>() * 4);
#pragma unroll
            for (int i = 0; i < 8; i++) lds_store(L.wa[i] + so, Aw[i]);
        }
        wave_lds_fence();                        // the patch words are complete; the J tiles reuse the I tiles' LDS
#pragma unroll
        for (int s = 0; s < kSlots; s++) {
            if (!((m_j >> (16 * s)) & 1ull)) continue;
            tile_store_j2<false>(lds + s * kTileDw2, rJ[s], q_dst, q_dc4, lane);
        }
        float A11, A12, A22, D;
        {
            float r11, r12, r22;
            chain_a(L, r11, r12, r22);
            // iA += A_buf[0] + A_buf[1] + A_buf[2] + A_buf[3] after the scalar tail went into iA
            A11 = combine_a(r11) * FLT_SCALE;
            A12 = combine_a(r12) * FLT_SCALE;
            A22 = combine_a(r22) * FLT_SCALE;
        }
        wave_lds_fence();                        // the iterations' b terms overwrite the patch words
        D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * kWin * kWin);
        const bool degenerate = minEig < 0.001f || D < 1.1920929e-07f;
        if (lvl_on && degenerate && level == 0) status = 0;
        lvl_on = lvl_on && !degenerate;
        D = 1.f / D;

        if (level > 0) request_I(level - 1);
        float pdx = 0.f, pdy = 0.f;
        bool it_on = lvl_on;
        for (int j = 0; j < kLkMaxIter; j++) {
            if (!__any(it_on)) break;
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (it_on && window_oob(inx, iny, w, h)) {
                if (level == 0) status = 0;
                it_on = false;
            }
            const PackedWeights wj = bilinear_weights(qx - (float)inx, qy - (float)iny);
            const uint32_t Wa = wj.Wa, Wb = wj.Wb;
            int cx = inx - tx0, cy = iny - ty0;
            const bool restage = it_on && ((unsigned)cx > (unsigned)(2 * kJMargin2) || (unsigned)cy > (unsigned)(2 * kJMargin2));
            if (restage) { tx0 = inx - kJMargin2; ty0 = iny - kJMargin2; cx = kJMargin2; cy = kJMargin2; }
            const int joff = ((int)__umul24((unsigned)cx, kCS2) + cy + (lane >> 4) * kTileDw2) * 4;
            const unsigned long long m_it = __ballot(it_on), m_rs = __ballot(restage);
            if (__builtin_expect(m_rs != 0, 0)) {     // a window drifted out of its tile
#pragma unroll
                for (int s = 0; s < kSlots; s++) {
                    if (!((m_rs >> (16 * s)) & 1ull)) continue;
                    const int tx0s = __builtin_amdgcn_readlane(tx0, 16 * s), ty0s = __builtin_amdgcn_readlane(ty0, 16 * s);
                    uint32_t r[3][2];
                    tile_loads(r, slotJ, slotJ + pitch, (uint32_t)(g.origin[level] + ty0s * pitch + tx0s), q_src, lane, kJPairs2 * 7);
                    tile_store_j2<true>(lds + s * kTileDw2, r, q_dst, q_dc4, lane);
                }
                wave_lds_fence();
            }
#pragma unroll
            for (int s = 0; s < kSlots; s++) {
                if (!((m_it >> (16 * s)) & 1ull)) continue;
                const int joffs = __builtin_amdgcn_readlane(joff, 16 * s);
                const uint32_t Was = __builtin_amdgcn_readlane(Wa, 16 * s), Wbs = __builtin_amdgcn_readlane(Wb, 16 * s);
                lds_cu32 *pj = (lds_cu32 *)(size_t)(lds_base + (uint32_t)joffs + L.qoff);
                uint32_t C[9];
#pragma unroll
                for (int k = 0; k < 9; k++) C[k] = pj[k * kCS2];
                float v[16];
                mismatch_slot8<LEGACY>(C, Was, Wbs, IvP[s], IxP[s], IyP[s], Ix4[s], Iy4[s], vround, v);
                const uint32_t so = (uint32_t)(s * stage_dw<LEGACY>() * 4);
#pragma unroll
                for (int t = 0; t < (LEGACY ? 16 : 10); t++) lds_store(L.wb[t] + so, __float_as_uint(v[t]));
            }
            wave_lds_fence();
            float b1f, b2f;
            {
                float r;
#ifdef SVO_LKS_EXPERIMENTS
                float guard = 0.f;
                if (EXP == 1) __builtin_amdgcn_s_setprio(3);                       // the chain wave first at the issue arbiter
                if (EXP == 2) r = chain_b_tree<false>(L, lane, guard);             // no serial chain at all: the bound
                else if (EXP == 3) {                                               // tree + guard max(P, N) = (sum |t| + |S|) / 2 < 2^24, else the chain
                    r = chain_b_tree<true>(L, lane, guard);
                    if (__builtin_expect(__any(guard + __builtin_fabsf(r) >= 33546240.f), 0)) r = chain_b<LEGACY>(L);
                } else r = chain_b<LEGACY>(L);
                if (EXP == 1) __builtin_amdgcn_s_setprio(0);
#else
                r = chain_b<LEGACY>(L);
#endif
                // bbuf = qb0 + qb1; ib1 += bbuf[0] + bbuf[2]; ib2 += bbuf[1] + bbuf[3]  (the tails are already in ib), on the
                // DPP network inside the row.  Positions: 0 tail x | 1..4 chains 0, 4, 2, 6 | 8 tail y | 9..12 chains 1, 5, 3, 7:
                // position 1 + 2 = bb0, 3 + 4 = bb2 (9.. : bb1, bb3), then their sum, then the tail at the head of the half row
                const float bb = r + row_shl<1>(r);
                const float u = bb + row_shl<2>(bb);
                const float f = r + row_shl<1>(u);
                b1f = row_first(f) * FLT_SCALE;
                b2f = row_first(row_shl<8>(f)) * FLT_SCALE;
            }
            wave_lds_fence();                    // the next iteration's terms overwrite these
            const float dlx = (A12 * b2f - A22 * b1f) * D;
            const float dly = (A12 * b1f - A11 * b2f) * D;
            const float dd = dlx * dlx + dly * dly;
            bool conv = dd <= 0.9999e-4f;
            if (__builtin_expect(__any(it_on && !conv && dd < 1.0001e-4f), 0)) {
                asm volatile("" ::: "memory");
                conv = (double)dlx * (double)dlx + (double)dly * (double)dly <= 0.01 * 0.01;
            }
            if (it_on) {
                qx += dlx; qy += dly;
                nx = qx + half; ny = qy + half;
                if (conv) it_on = false;
                else if (j > 0 && fabsf(dlx + pdx) <= 0.01f && fabsf(dly + pdy) <= 0.01f) {
                    nx -= dlx * 0.5f; ny -= dly * 0.5f;
                    it_on = false;
                }
                pdx = dlx; pdy = dly;
            }
        }
        if (live && status && level == 0) {
            int fx = cv_floor(nx - half), fy = cv_floor(ny - half);
            if (window_oob(fx, fy, w, h)) status = 0;
        }
    }
    outPt = make_float2(nx, ny);
}

// Grid: ONE wave per workgroup, a.gx WAVES per batch item walking the item's points in strides of a.gx * 4 slots;
// the same XCD-aware item mapping as lk_kernel (consecutive workgroup ids go round the 8 XCDs).
template <bool LEGACY, int EXP = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void lk_sse2_kernel(LkArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LEGACY ? kLdsDwLegacy : kLdsDwSse2];
    const int n_aware = (a.batch & ~7) * a.gx;
    int b, wv;
    if ((int)blockIdx.x < n_aware) {
        const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
        b = (slot_id / a.gx) * 8 + xcd; wv = slot_id % a.gx;
    } else {
        const int r = blockIdx.x - n_aware;
        b = (a.batch & ~7) + r / a.gx; wv = r % a.gx;
    }
    const int lane = threadIdx.x & 63;
    const int slot = lane >> 4;
    int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    n = min(n, a.cap);
    const Sse2Lane L = make_lane<LEGACY>(lane, (uint32_t)(size_t)(lds_cu32 *)lds, a.accum == 2);
    int spw = kSlots;
    if (a.spread) spw = min(kSlots, max(1, (n + a.gx - 1) / a.gx));
    for (int first = wv * spw; first < n; first += a.gx * spw) {
        const int idx = first + slot;
        const bool valid = slot < spw && idx < n;
        const bool writer = valid && lane == 16 * slot;
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
            lk_call4_sse2<LEGACY, EXP>(a.g, sI, sJ, cur, nxt, st, live, lds, lane, L);
            if (writer && live) {
                a.pts_out[c][po] = nxt;
                a.status[c][po] = (uint8_t)st;
            }
            // Tracking::deleteBadmatchFeatures terms, as in lk_kernel (src/tracking.cpp:619-660)
            if (live) {
                outside = outside || nxt.x < 0 || nxt.y < 0;
                bad = bad || st == 0;
                if (c == 0 || c == 2) noepi = noepi || (double)fabsf(prev_y - nxt.y) > a.match_err;
                prev_y = nxt.y;
                cur = nxt;
            }
            if (a.ncalls == 4 && (outside || bad || noepi)) live = false;
            if (!__any(live)) break;
        }
        if (a.ncalls == 4 && writer) a.keep[po] = !(outside || bad || noepi);
        wave_lds_fence();
    }
}

void launch_lk_sse2(const LkArgs &a0, int batch, int max_pts, hipStream_t st)
{
    if (max_pts <= 0 || batch <= 0) return;
    const int waves = (max_pts + kSlots - 1) / kSlots;      // one pass over the capacity
    LkArgs a = a0;
    a.gx = waves < 768 ? waves : 768;
    a.spread = 0;
    if (batch < 4) {                             // latency shape: eight single-wave workgroups per CU are resident
        const int room = 2048 / batch;
        a.gx = max_pts < room ? max_pts : room;
        a.spread = 1;
    }
    a.batch = batch;
    if (a.accum == 3) {
        if (batch < 4) {                          // six legacy workgroups per CU
            const int room = 1536 / batch;
            a.gx = max_pts < room ? max_pts : room;
        }
        hipLaunchKernelGGL(lk_sse2_kernel<true>, dim3(batch * a.gx), dim3(64), 0, st, a);
    } else {
#ifdef SVO_LKS_EXPERIMENTS
        static const int exp_variant = getenv("SVO_LKS_EXP") ? atoi(getenv("SVO_LKS_EXP")) : 0;
        if (exp_variant == 1) hipLaunchKernelGGL((lk_sse2_kernel<false, 1>), dim3(batch * a.gx), dim3(64), 0, st, a);
        else if (exp_variant == 2) hipLaunchKernelGGL((lk_sse2_kernel<false, 2>), dim3(batch * a.gx), dim3(64), 0, st, a);
        else if (exp_variant == 3) hipLaunchKernelGGL((lk_sse2_kernel<false, 3>), dim3(batch * a.gx), dim3(64), 0, st, a);
        else
#endif
        hipLaunchKernelGGL(lk_sse2_kernel<false>, dim3(batch * a.gx), dim3(64), 0, st, a);
    }
}

}  // namespace svo
