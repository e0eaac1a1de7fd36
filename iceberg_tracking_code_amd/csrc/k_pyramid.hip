// k_pyramid.hip -- the Gaussian pyramid of a frame in ONE launch (up to three pyrDown levels per launch) for gfx950.
//
// Replaces the per-level pyrDown launches inside cv2.calcOpticalFlowPyrLK -> buildOpticalFlowPyramid
// (s1_lucaskanade_tracking.py:323,326; arithmetic restated from OpenCV's pyrDown, SURVEY.md A.2/A.3: separable
// [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101 at every level, dst = ((w+1)/2, (h+1)/2)).
//
// Why one launch: level l+1 needs level l complete, so three launches are three dependent passes, and the upper two
// have too few pixels to fill the chip: each lasts one workgroup's latency (~8 us) whatever its size -- 33 us per
// 12 MP frame for 19.7 MB of traffic, 7 % of the HBM rate.  Here a workgroup owns a 128x128 tile of level 0 and builds
// everything above it: the 64x64 tile of level 1, 32x32 of level 2, 16x16 of level 3.  The level-l values a tile needs
// from its neighbours (2 px each side, doubling downwards: 14 px of level 0) are recomputed rather than exchanged:
// 35 % more level-0 bytes through L2, no inter-workgroup dependency, every HBM byte of level 0 read once.
//   stage 1  the level-0 region (149 rows x 156 B) into LDS as aligned dwords, all loads of a thread in flight at once
//   stage 2  level 1 region (73 x 75) from LDS into LDS; the owned 64x64 block goes out as whole dwords
//   stage 3  level 2 region (35 x 35) likewise, stage 4  level 3 (16 x 16)
// A task = one output column x a run of CH output rows: the horizontal 5-tap sum of a source row is ONE
// v_dot4_u32_u8 (weights 1 4 6 4 on four bytes cut out of two LDS dwords by v_alignbyte) + the fifth byte, and is
// shared by the 2-3 output rows that use the row: ~13 vector instructions per output pixel.
// Borders: a tile at the frame edge completes every region before it is read -- the two positions outside each image
// edge (all that the taps of an in-image output can reach) are copied, inside LDS, from their mirror positions
// (BORDER_REFLECT_101 of that level's own image).  The 5x5 arithmetic itself never looks at a coordinate: one code
// path for every tile; outputs outside the image are computed on whatever the region holds and never stored.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "icelk_internal.h"

namespace icelk {

namespace {

__device__ __forceinline__ int reflect101(int p, int n)
{
    // every coordinate a tile asks for lies within [-20, n + 156): one reflection does it unless the level is smaller
    // than that (n is the same for every lane: a scalar branch); the loop handles any p and n
    if (n >= 256) {
        p = p < 0 ? -p : p;
        return p >= n ? 2 * n - 2 - p : p;
    }
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Geometry of a workgroup that owns a T0 x T0 tile of level 0 (T0 = 128 with 256 threads: 156 x 149 B region, 42 % halo;
// T0 = 64 with ONE wave: 92 x 85, 91 % halo -- more bytes through L2 and more arithmetic, but a one-wave workgroup fits
// into the slot a single retiring tracker wave leaves, so a pyramid built beside a tracker launch is not held up until
// that launch has drained; a 4-wave workgroup never finds four free slots on one CU while the tracker has workgroups
// pending).  Region origins are relative to the tile origin of that level; columns are padded to dword multiples so
// that the owned block starts on a dword.
template <int T0_, int NT_>
struct Geo {
    static constexpr int T0 = T0_, NT = NT_;
    static constexpr int L0_OX = -20, L0_OY = -14, L0_W = T0 + 28, L0_H = T0 + 21;   // needs [-18,T0+6] x [-14,T0+6]
    static constexpr int L1_OX = -8, L1_OY = -6, L1_W = T0 / 2 + 12, L1_H = T0 / 2 + 9;   // needs [-6,T0/2+2]: columns 2.. of the region
    static constexpr int L2_OX = -4, L2_OY = -2, L2_W = T0 / 4 + 8, L2_H = T0 / 4 + 3;    // needs [-2,T0/4]: columns 2.. of the region
    static constexpr int L3_W = T0 / 8, L3_H = T0 / 8;
    // rows per task of level 2: 35 rows x 35 columns are 245 tasks of 5 rows for 256 threads (runs of 8 left a third of
    // the threads idle behind 19-row tasks); the one-wave geometry (19 x 19) stays one round of 57 tasks with runs of 8
    static constexpr int L2_CH = T0 == 128 ? 5 : 8;
    static constexpr int L1_CH = T0 == 128 ? 6 : 8;   // rows per task of level 1 (column pairs), see the call
    // LDS row pitch of the level-0 region: a multiple of 16 B, so that a tile inside the frame is fetched and stored as
    // 16-byte pieces (global_load_dwordx4 -> ds_write_b128: a quarter of the load and store instructions)
    static constexpr int L0_P = (L0_W + 15) & ~15;
    static constexpr int L0_BYTES = L0_P * L0_H, L1_BYTES = L1_W * L1_H, L2_BYTES = L2_W * L2_H, L3_BYTES = L3_W * L3_H;
    static constexpr int LDS_BYTES = ((L0_BYTES + 15) & ~15) + ((L1_BYTES + 15) & ~15) + ((L2_BYTES + 15) & ~15) + ((L3_BYTES + 15) & ~15);
};

struct LevelIO {
    uint8_t* ptr;
    int w, h, pitch;
};

// One task of a level: output column `ox` (region index), output rows oy0 .. oy0+CH-1 (region indices) of a
// destination region, from a source region in LDS.
//   src col of tap k = 2*ox + scol + k,  src row of tap k of output row oy = 2*oy + srow + k   (region indices)
// interior: every tap is inside the source region as it stands.  Otherwise the taps are reflected on the GLOBAL
// coordinates of the source level (gsx0, gsy0 = global coordinates of source region index 0).
template <int CH>
__device__ __forceinline__ void pyr_task(const uint8_t* __restrict__ src, int spitch, int scol, uint8_t* __restrict__ dst,
                                         int dpitch, int ox, int oy0)
{
    constexpr int NR = 2 * CH + 3;
    int hs[NR];
    const int c0 = 2 * ox + scol;
#pragma unroll
    for (int i = 0; i < NR; i++) {
        // bytes c0 .. c0+4 of the row out of the two dwords that hold them (c0 is even: byte offset 0 or 2): the four
        // weighted ones cut out by v_alignbyte, summed by one v_dot4_u32_u8 with the fifth as its addend
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (2 * oy0 + i) * spitch + (c0 & ~3));
        const uint32_t d0 = p[0], d1 = p[1];
        const uint32_t four = __builtin_amdgcn_alignbyte(d1, d0, c0 & 3);
        const uint32_t fifth = (c0 & 2) ? ((d1 >> 16) & 255u) : (d1 & 255u);
        hs[i] = (int)__builtin_amdgcn_udot4(four, 0x04060401u, fifth, false);
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int v = hs[2 * j] + hs[2 * j + 4] + 4 * (hs[2 * j + 1] + hs[2 * j + 3]) + 6 * hs[2 * j + 2];
        dst[(oy0 + j) * dpitch + ox] = (uint8_t)((v + 128) >> 8);
    }
}

// One level of the tile: NCOL x ROWS outputs of the destination region (columns from column FIRST_COL of the region),
// runs of CH rows per task; the ROWS % CH rows left over are tasks of their own (a run of CH with one live row would
// cost as much as a full one).  Region index (ox, oy) <-> source region: column 2*ox + scol, row 2*oy.
// CH0 .. CH1-1 = the runs of CH rows this call forms (all of them by default); the leftover rows go with the last run
template <int NT, int CH, int NCOL, int ROWS, int FIRST_COL, int CH0 = 0, int CH1 = ROWS / CH>
__device__ __forceinline__ void pyr_level(const uint8_t* __restrict__ src, int spitch, int scol, uint8_t* __restrict__ dst,
                                          int dpitch, int tid)
{
    constexpr int NFULL = ROWS / CH, REM = ROWS - NFULL * CH;
    for (int t = tid; t < NCOL * (CH1 - CH0); t += NT) {
        const int ch = CH0 + t / NCOL;
        pyr_task<CH>(src, spitch, scol, dst, dpitch, FIRST_COL + (t - (ch - CH0) * NCOL), ch * CH);
    }
    if constexpr (REM > 0 && CH1 == NFULL) {
        // the leftover rows: threads from the far end of the workgroup, so that they fall into the partly filled last
        // round of the loop above rather than into a round of their own
        for (int t = NT - 1 - tid; t < NCOL; t += NT) pyr_task<REM>(src, spitch, scol, dst, dpitch, FIRST_COL + t, NFULL * CH);
    }
}

// The same for TWO adjacent output columns ox, ox + 1 whose first source byte c0 = 2*ox + scol is a multiple of 4
// (round 4): the seven source bytes of a row are the two aligned dwords at c0, column ox sums the first dword as it
// stands + byte 4, column ox + 1 the dword cut out at byte 2 + byte 6 -- five vector instructions per source row for
// two columns (no lane-dependent shift) where the one-column task needs four for one.
template <int CH>
__device__ __forceinline__ void pyr_task_pair(const uint8_t* __restrict__ src, int spitch, int scol, uint8_t* __restrict__ dst,
                                              int dpitch, int ox, int oy0)
{
    constexpr int NR = 2 * CH + 3;
    int ha[NR], hb[NR];
    const int c0 = 2 * ox + scol;
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (2 * oy0 + i) * spitch + c0);
        const uint32_t d0 = p[0], d1 = p[1];
        ha[i] = (int)__builtin_amdgcn_udot4(d0, 0x04060401u, d1 & 255u, false);
        hb[i] = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), 0x04060401u, (d1 >> 16) & 255u, false);
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const int va = ha[2 * j] + ha[2 * j + 4] + 4 * (ha[2 * j + 1] + ha[2 * j + 3]) + 6 * ha[2 * j + 2];
        const int vb = hb[2 * j] + hb[2 * j + 4] + 4 * (hb[2 * j + 1] + hb[2 * j + 3]) + 6 * hb[2 * j + 2];
        dst[(oy0 + j) * dpitch + ox] = (uint8_t)((va + 128) >> 8);
        dst[(oy0 + j) * dpitch + ox + 1] = (uint8_t)((vb + 128) >> 8);
    }
}

// One level of the tile out of column pairs: NPAIR pairs from column FIRST_COL (2*FIRST_COL + scol a multiple of 4) x ROWS
// rows, runs of CH rows per task, the leftover rows as tasks of their own on the far threads (as in pyr_level)
template <int NT, int CH, int NPAIR, int ROWS, int FIRST_COL>
__device__ __forceinline__ void pyr_level_pairs(const uint8_t* __restrict__ src, int spitch, int scol, uint8_t* __restrict__ dst,
                                                int dpitch, int tid)
{
    constexpr int NFULL = ROWS / CH, REM = ROWS - NFULL * CH;
    for (int t = tid; t < NPAIR * NFULL; t += NT) {
        const int ch = t / NPAIR;
        pyr_task_pair<CH>(src, spitch, scol, dst, dpitch, FIRST_COL + 2 * (t - ch * NPAIR), ch * CH);
    }
    if constexpr (REM > 0) {
        for (int t = NT - 1 - tid; t < NPAIR; t += NT) pyr_task_pair<REM>(src, spitch, scol, dst, dpitch, FIRST_COL + 2 * t, NFULL * CH);
    }
}

// A tile at the frame edge: the two positions next to each image edge (all a 5-tap filter centred inside the image can
// reach) take the value of their mirror positions (BORDER_REFLECT_101), which the same region holds.  Round 4: ONE pass
// over the sides the region really touches (rounds 1-3: all four sides of every edge tile, columns first, a barrier, then
// rows -- five rounds of mostly idle threads and two barriers per level, 1 500 cycles per level on every edge tile while
// all other workgroups wait for the launch to end).  A position outside the image in x, in y or in both reads the
// position INSIDE the image that both reflections lead to; in-image positions are never written here, so nothing read
// in this pass is written in it, and a corner position reached from both lists gets the same value twice.
// (gx0, gy0) = global coordinates of region (0, 0).  ROWS_TOO = false: the rows outside the image hold their mirror rows
// already (level 0: the loads fetched them from there), only the columns are left to do.
template <int NT, int COLS, int ROWS, bool ROWS_TOO>
__device__ __forceinline__ void fill_edges(uint8_t* __restrict__ reg, int pitch, int gx0, int gy0, int w, int h, int tid)
{
    const bool left = gx0 < 0, right = gx0 + COLS > w, top = ROWS_TOO && gy0 < 0, bottom = ROWS_TOO && gy0 + ROWS > h;
    const int n_col_items = ((left ? 2 : 0) + (right ? 2 : 0)) * ROWS, n_row_items = ((top ? 2 : 0) + (bottom ? 2 : 0)) * COLS;
    for (int i = tid; i < n_col_items + n_row_items; i += NT) {
        int r, c;
        if (i < n_col_items) {
            const int k = i / ROWS;                     // 0, 1: the first side that is there; 2, 3: the second
            r = i - k * ROWS;
            c = ((left && k < 2) ? k - 2 : w + (k & 1)) - gx0;
        } else {
            const int j = i - n_col_items, k = j / COLS;
            c = j - k * COLS;
            r = ((top && k < 2) ? k - 2 : h + (k & 1)) - gy0;
        }
        const int sc = reflect101(gx0 + c, w) - gx0, sr = ROWS_TOO ? reflect101(gy0 + r, h) - gy0 : r;
        if ((unsigned)c < (unsigned)COLS && (unsigned)r < (unsigned)ROWS && (unsigned)sc < (unsigned)COLS &&
            (unsigned)sr < (unsigned)ROWS)
            reg[r * pitch + c] = reg[sr * pitch + sc];
    }
    __syncthreads();
}

// the owned block of a region goes to global memory in pieces of PB = 16 bytes (8 for a block 8 bytes wide): rows
// [ry, ry+ROWS) x byte columns [rx, rx + BYTES) of the region -> (gx0, gy0) of the level.  gx0 is a multiple of BYTES, the
// level starts on a 256-B boundary and its pitch is a multiple of 64, so a piece is aligned and one that starts inside
// the image ends inside its row's pitch (what it writes behind column w - 1 is padding nothing reads).  Round 4: one
// 16-byte store per thread for the 64 x 64 block of level 1 where there were four dword stores with an LDS read between them.
template <int NT, int ROWS, int BYTES>
__device__ __forceinline__ void copy_out(const uint8_t* __restrict__ reg, int rpitch, int rx, int ry, const LevelIO& L, int gx0,
                                         int gy0, int tid)
{
    constexpr int PB = BYTES >= 16 ? 16 : 8, PPR = BYTES / PB;
    static_assert(BYTES % PB == 0, "block width");
    for (int i = tid; i < ROWS * PPR; i += NT) {
        const int r = i / PPR, c = i - r * PPR;
        const int gy = gy0 + r, gx = gx0 + PB * c;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(reg + (ry + r) * rpitch + rx + PB * c);   // 4-byte aligned
        if (gy < L.h && gx < L.w) {
            uint8_t* g = L.ptr + (size_t)gy * L.pitch + gx;
            // write-through stores (sc0 sc1): the 3.9 MB a launch writes go to memory as they are written instead of waiting in
            // the L2s for the write-back at the end of the kernel: 7.46 -> 7.30 us, one-wave form 9.84 -> 9.51.  (Nontemporal
            // stores, which also bypass the L2's write combining, measured 7.8.)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if constexpr (PB == 16) {
                const u32x4 v = {q[0], q[1], q[2], q[3]};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(g), "v"(v) : "memory");
            } else {
                const u32x2 v = {q[0], q[1]};
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(g), "v"(v) : "memory");
            }
        }
    }
}

template <int NL, int T0, int NT>
__global__ __launch_bounds__(NT) void k_pyramid(LevelIO S, LevelIO D1, LevelIO D2, LevelIO D3, unsigned long long* stamps, int tiles_x,
                                                 int tiles_y)
{
    // Which tile a workgroup takes (round 4).  Two things decide it:
    //  * every workgroup of the launch is resident at once and all of them ask for their level-0 regions in the same
    //    microsecond, so a workgroup's load stage is as long as its place in the memory system's queue, and the launch
    //    lasts as long as its slowest workgroup: the tiles on the frame edge (which also fill their borders) go FIRST
    //    (dispatched last, the bottom row had 13 000 cycles of load stage against 7 800 inside the frame);
    //  * workgroup b runs on XCD b % 8, and every XCD has an L2 of its own: tiles that share a halo belong on ONE XCD, or
    //    the 14-px halo (49 % of the tile's own bytes) comes out of HBM once per neighbour.  XCD k takes the k-th eighth
    //    of the tiles in raster order -- a band of the frame -- and inside its band the left and right columns first,
    //    then the rest row by row: downwards in the upper half of the frame, upwards in the lower half, so that the top
    //    and the bottom row of the frame are the first rows of their bands.
    int tbx, tby;
    {
        const int gx = tiles_x, n = tiles_x * tiles_y, per = (n + 7) / 8;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int a = xcd * per, e = a + per < n ? a + per : n;
        if (a + slot >= e) return;                                   // the last bands of a grid that does not divide by 8
        const bool up = xcd >= 4;
        int idx;
        if (gx < 3) {
            idx = up ? e - 1 - slot : a + slot;
        } else {
            const int row_a = a / gx, row_e = e / gx;
            const int n_left = (e + gx - 1) / gx - (a + gx - 1) / gx;   // raster indices = 0 (mod gx) in [a, e)
            const int n_right = row_e - row_a;                          // ... = gx - 1 (mod gx)
            if (slot < n_left) idx = ((a + gx - 1) / gx + slot) * gx;
            else if (slot < n_left + n_right) idx = (row_a + slot - n_left) * gx + gx - 1;
            else {
                // the k-th tile of the band that is in neither column, counted in a raster of the gx - 2 inner columns
                const int inner = gx - 2, k = slot - n_left - n_right;
                auto inner_before = [&](int i) {
                    const int c = i % gx - 1;
                    return (i / gx) * inner + (c < 0 ? 0 : c > inner ? inner : c);
                };
                const int first = inner_before(a), count = inner_before(e) - first;
                const int rank = first + (up ? count - 1 - k : k);
                idx = (rank / inner) * gx + 1 + rank % inner;
            }
        }
        tby = idx / gx;
        tbx = idx - tby * gx;
    }
#define STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[8 * (tby * tiles_x + tbx) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);
    using G = Geo<T0, NT>;
    constexpr int L0_P = G::L0_P;
    constexpr int L0_OX = G::L0_OX, L0_OY = G::L0_OY, L0_W = G::L0_W, L0_H = G::L0_H, L1_OX = G::L1_OX, L1_OY = G::L1_OY,
                  L1_W = G::L1_W, L1_H = G::L1_H, L2_OX = G::L2_OX, L2_OY = G::L2_OY, L2_W = G::L2_W, L2_H = G::L2_H,
                  L3_W = G::L3_W, L3_H = G::L3_H;
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::LDS_BYTES];
    uint8_t* R0 = lds;
    uint8_t* R1 = R0 + ((G::L0_BYTES + 15) & ~15);
    uint8_t* R2 = R1 + ((G::L1_BYTES + 15) & ~15);
    uint8_t* R3 = R2 + ((G::L2_BYTES + 15) & ~15);
    const int tid = threadIdx.x;
    const int x0 = tbx * T0, y0 = tby * T0;          // level-0 tile origin
    const int x1 = x0 / 2, y1 = y0 / 2, x2 = x0 / 4, y2 = y0 / 4, x3 = x0 / 8, y3 = y0 / 8;
    // every tap of every level inside its region as it stands <=> the level-0 region lies inside the image (the
    // regions of the upper levels are its images)
    const bool interior = x0 + L0_OX >= 0 && y0 + L0_OY >= 0 && x0 + L0_OX + L0_W <= S.w && y0 + L0_OY + L0_H <= S.h;

    // ---- stage 1: level-0 region into LDS ---------------------------------------------------------------------------
    // Aligned dwords; a dword is fetched iff its first column and its row are inside the image -- the region origin is
    // 4-aligned, so no dword straddles the LEFT edge, and one that straddles the right edge reads into the row padding
    // (pitch is a multiple of 64), bytes the edge fill below overwrites or nothing ever looks at.  No byte path, no
    // branch: every load of the thread is in flight before the first LDS write waits for one.
    if (interior) {
        // 16 bytes per load and per LDS store; the last piece of a row runs up to 12 B past the region: inside the frame's
        // row pitch (the region ends inside the frame, the pitch is a multiple of 64), and bytes nothing reads
        constexpr int NQ = L0_P / 16, N = (NQ * L0_H + NT - 1) / NT;
        uint4 v[N];
        const uint8_t* base = S.ptr + (size_t)(y0 + L0_OY) * S.pitch + (x0 + L0_OX);
#pragma unroll
        for (int m = 0; m < N; m++) {
            int i = tid + NT * m;
            i = i < NQ * L0_H ? i : NQ * L0_H - 1;      // surplus threads of the last round fetch the last piece again
            const int r = i / NQ, c = i - r * NQ;
            v[m] = *reinterpret_cast<const uint4*>(base + (size_t)r * S.pitch + 16 * c);
        }
#pragma unroll
        for (int m = 0; m < N; m++) {
            int i = tid + NT * m;
            i = i < NQ * L0_H ? i : NQ * L0_H - 1;
            reinterpret_cast<uint4*>(R0)[i] = v[m];
        }
        // (Committing the first two thirds of the rounds, forming the level-1 runs that read nothing below them, then the
        // rest -- so that the last loads are still in flight during the first runs -- was measured and is slower: 12.1 ->
        // 13.1 us, 18.8 -> 22.0 us for the one-wave geometry; the extra barrier and the extra partly filled round of tasks
        // cost more than the overlap returns.)
    } else {
        // A tile at the frame edge (round 4): the same 16-byte pieces, each fetched iff its row lies inside the image and it
        // overlaps the image's columns.  (Rounds 1-3 fetched such tiles dword by dword, 23 bounds-checked loads per thread
        // with a division each, and -- all workgroups of a launch being resident at once -- the launch lasted as long as its
        // slowest workgroup: 25 000 cycles for a tile on the bottom edge against 16 000 for one inside the frame,
        // tools/pyr_stamps_run.sh.)  A piece that sticks out on the right reads up to 12 bytes of the row's padding (the
        // pitch is a multiple of 64) or of the next row -- inside the allocation, see layout_ok -- which the edge fill
        // overwrites or nothing looks at.  The one piece that straddles the LEFT edge (region origin -20: image columns
        // -4 .. 11) is fetched from column 0 instead and lands 4 bytes further on, as dwords.
        constexpr int NQ = L0_P / 16, N = (NQ * L0_H + NT - 1) / NT;
        uint4 v[N];
        const int gx0 = x0 + L0_OX, gy0 = y0 + L0_OY;
#pragma unroll
        for (int m = 0; m < N; m++) {
            int i = tid + NT * m;
            i = i < NQ * L0_H ? i : NQ * L0_H - 1;
            const int r = i / NQ, c = i - r * NQ;
            const int gy = gy0 + r, gx = gx0 + 16 * c;
            const bool ok = gx < S.w && gx + 15 >= 0;
            const int xl = gx < 0 ? 0 : gx;
            // a row outside the image is fetched from its mirror row: the top and bottom borders of level 0 cost nothing
            v[m] = ok ? *reinterpret_cast<const uint4*>(S.ptr + (size_t)reflect101(gy, S.h) * S.pitch + xl) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int m = 0; m < N; m++) {
            int i = tid + NT * m;
            i = i < NQ * L0_H ? i : NQ * L0_H - 1;
            const int r = i / NQ, c = i - r * NQ;
            const int gx = gx0 + 16 * c;
            if (gx < 0 && gx + 15 >= 0) {
                uint32_t* q = reinterpret_cast<uint32_t*>(R0 + r * L0_P + (0 - gx0));   // where image column 0 sits
                q[0] = v[m].x; q[1] = v[m].y; q[2] = v[m].z; q[3] = v[m].w;
            } else {
                reinterpret_cast<uint4*>(R0)[i] = v[m];
            }
        }
    }
    // ---- stages 2-4: level 1 (region columns 2 .. W-2, every row), level 2 (columns 2 .. T0/4+4), level 3 ----
    // region index (ox, oy) of a level <-> source region: column 2*ox + scol, row 2*oy, with scol = +2, -2, +2
    __syncthreads();
    if (!interior) fill_edges<NT, L0_W, L0_H, false>(R0, L0_P, x0 + L0_OX, y0 + L0_OY, S.w, S.h, tid);
    STAMP(1);
    // level 1 out of column pairs from column 1 (source byte 4): columns 1 .. L1_W - 2, of which 2 .. are needed.
    // 128-px form: 37 pairs x (12 runs of 6 rows + 1 row) = 481 tasks, two rounds of 256 threads; one-wave form: 21 pairs x
    // (5 runs of 8 + 1) = 126 tasks, two rounds of 64
    pyr_level_pairs<NT, G::L1_CH, (L1_W - 2) / 2, L1_H, 1>(R0, L0_P, 2, R1, L1_W, tid);
    __syncthreads();
    STAMP(2);
    copy_out<NT, T0 / 2, T0 / 2>(R1, L1_W, -L1_OX, -L1_OY, D1, x1, y1, tid);
    STAMP(3);
    if (NL == 1) return;
    if (!interior) fill_edges<NT, L1_W, L1_H, true>(R1, L1_W, x1 + L1_OX, y1 + L1_OY, D1.w, D1.h, tid);
    pyr_level<NT, G::L2_CH, L2_W - 5, L2_H, 2>(R1, L1_W, -2, R2, L2_W, tid);
    __syncthreads();
    STAMP(4);
    copy_out<NT, T0 / 4, T0 / 4>(R2, L2_W, -L2_OX, -L2_OY, D2, x2, y2, tid);
    if (NL == 2) return;
    if (!interior) fill_edges<NT, L2_W, L2_H, true>(R2, L2_W, x2 + L2_OX, y2 + L2_OY, D2.w, D2.h, tid);
    pyr_level<NT, 2, L3_W, L3_H, 0>(R2, L2_W, 2, R3, L3_W, tid);
    __syncthreads();
    copy_out<NT, T0 / 8, T0 / 8>(R3, L3_W, 0, 0, D3, x3, y3, tid);
    STAMP(5);
}

LevelIO io_of(const Level& L)
{
    LevelIO o;
    o.ptr = L.ptr;
    o.w = L.w;
    o.h = L.h;
    o.pitch = L.pitch;
    return o;
}

}  // namespace

template <int T0, int NT>
static void launch_geo(hipStream_t s, const Level* lv, int first, int n)
{
    const Level& S = lv[first];
    const int tiles_x = (S.w + T0 - 1) / T0, tiles_y = (S.h + T0 - 1) / T0;
    dim3 grid(8 * ((tiles_x * tiles_y + 7) / 8));   // eight bands of equal length, see the kernel
    const LevelIO src = io_of(S), d1 = io_of(lv[first + 1]);
    const LevelIO d2 = n >= 2 ? io_of(lv[first + 2]) : d1, d3 = n >= 3 ? io_of(lv[first + 3]) : d1;
    // diagnostics: ICELK_PYR_STAMPS=<file> records s_memtime at the stage boundaries of every workgroup of each launch
    static unsigned long long* d_st = nullptr;
    static const char* st_path = getenv("ICELK_PYR_STAMPS");
    const size_t nst = 8 * (size_t)grid.x;
    if (st_path && !d_st) hipMalloc(reinterpret_cast<void**>(&d_st), 8 * 8 * 65536);
    if (n == 1) hipLaunchKernelGGL((k_pyramid<1, T0, NT>), grid, dim3(NT), 0, s, src, d1, d2, d3, d_st, tiles_x, tiles_y);
    else if (n == 2) hipLaunchKernelGGL((k_pyramid<2, T0, NT>), grid, dim3(NT), 0, s, src, d1, d2, d3, d_st, tiles_x, tiles_y);
    else hipLaunchKernelGGL((k_pyramid<3, T0, NT>), grid, dim3(NT), 0, s, src, d1, d2, d3, d_st, tiles_x, tiles_y);
    if (d_st && nst <= 8 * 65536) {
        std::vector<unsigned long long> hst(nst);
        hipStreamSynchronize(s);
        hipMemcpy(hst.data(), d_st, nst * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(st_path, "wb")) { fwrite(hst.data(), 8, nst, f); fclose(f); }
    }
}

// Builds lv[first+1 .. first+n] (n = 1..3) from lv[first] in one launch.  `one_wave`: the 64 x 64 / one-wave geometry,
// for a pyramid that is built beside a tracker launch (see Geo).
void launch_pyramid_fused(hipStream_t s, const Level* lv, int first, int n, bool one_wave)
{
    if (one_wave) launch_geo<64, 64>(s, lv, first, n);
    else launch_geo<128, 256>(s, lv, first, n);
}

}  // namespace icelk
