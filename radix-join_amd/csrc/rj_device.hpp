// rj_device.hpp — POD structs and tuning constants shared by the host executor
// and the gfx950 kernels.  Everything here is written for MI355X only: wave64,
// 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once
#include <stdint.h>

namespace rj {

// ---- Page geometry (reference include/plan.h:54, src/build_table.cpp:488,531)
constexpr uint32_t PAGE_BYTES = 8192;
constexpr uint32_t ROWS32 = 1984;  // rows of a full non-NULL INT32 page
constexpr uint32_t ROWS64 = 1007;  // rows of a full non-NULL INT64/FP64 page
constexpr uint32_t HDR32 = 4;      // first value offset, INT32
constexpr uint32_t HDR64 = 8;      // first value offset, INT64/FP64

// ---- Columns as the kernels see them
enum ColKind : int32_t {
    COL_NONE  = 0,
    COL_PAGED = 1,  // "regular" page images: every page but the last holds ROWS32/ROWS64
                    // non-NULL values, so row r lives at page r / ROWS, slot r % ROWS
    COL_DENSE = 2,  // flat array of 4- or 8-byte values (+ optional validity bytes)
    COL_IOTA  = 3   // value(row) = row: the row-id column of a base table (late
                    // materialisation of VARCHAR)
};

struct ColRef {
    const uint8_t* ptr;
    const uint8_t* valid;  // COL_DENSE only; nullptr = all rows valid
    int32_t        kind;
    int32_t        width;  // 4 or 8
};

// What travels with the key through the radix passes.
enum CarryMode : int32_t {
    CARRY_NONE   = 0,
    CARRY_ROWIDX = 1,  // row index into the child relation (generic path: gather later)
    CARRY_COLUMN = 2,  // the single payload column itself (direct path: no gather)
    CARRY_WIDE   = 3   // several payload columns (and/or a validity word) packed into 2 or 3 carry
                       // words, see TupleSrc::wide: the values themselves travel, nothing is
                       // gathered afterwards and nothing refers to a row of this rank
};

// Layout of a wide carry.  The executor orders the carried columns so that two loaders suffice:
enum WideLayout : int32_t {
    WIDE_NONE = 0,  // one column (carry), 32 or 64 bits
    WIDE_32S  = 1,  // two or three 32-bit columns: carry, carry2 [, carry3]
    WIDE_64_32 = 2  // a 64-bit column (carry) followed by a 32-bit one (carry2)
};

struct TupleSrc {
    ColRef   key;
    ColRef   carry;
    uint32_t n_rows;
    int32_t  carry_mode;
    int32_t  key_f64;    // FP64 key: compared by bit pattern, NaN never matches
    int32_t  prehashed;  // key column already holds hashed keys (sharded stage B)
    int32_t  wide;       // WideLayout (CARRY_WIDE)
    int32_t  pad;
    ColRef   carry2, carry3;  // further columns of a wide carry
};

// Partitioned tuples are SoA arrays of 32-bit words (or, for one key word + one carry word in
// fine-histogram plans, ONE array of 8-byte {word 0, carry} pairs — "packed"):
//   word 0            = hashed key (low 32 bits); radix digits and slot bits come from it
//   word 1 (KW == 2)  = high 32 bits of the 64-bit hashed key
//   following words   = carry (0, 1 or 2 words)
constexpr int MAX_WORDS = 4;
struct Words {
    uint32_t* w[MAX_WORDS];
};

// ---- Radix partition pass --------------------------------------------------
// One workgroup sorts a tile of PT_TILE tuples by digit in LDS and writes each
// digit's run contiguously (software write-combining), see rj_kernels.hip.
// Geometry is overridable at build time (-DRJ_PT_THREADS=...) for tuning runs.
#ifndef RJ_PT_THREADS
#define RJ_PT_THREADS 1024
#endif
#ifndef RJ_PT_ITEMS
#define RJ_PT_ITEMS 16
#endif
constexpr int PT_THREADS = RJ_PT_THREADS;         // waves = PT_THREADS / 64
constexpr int PT_ITEMS   = RJ_PT_ITEMS;           // tuples per thread per tile
constexpr int PT_TILE    = PT_THREADS * PT_ITEMS; // 16384 tuples = 64 KiB of LDS staging; longer
                                                  // digit runs per tile = fewer partial HBM lines
constexpr uint32_t PT_ALL_ITEMS = PT_ITEMS >= 32 ? 0xffffffffu : ((1u << (PT_ITEMS & 31)) - 1u);  // one bit per item
// what the kernels can fan out per pass; -DRJ_PT_CAPBITS=10 builds the 1024-digit variant that
// RJ_TUNE_P1_BITS=8 / 10 needs to cut 18 bits 8+10 / 10+8 (profiles/r03_r_bits_8_10_ab.log: no gain)
#ifndef RJ_PT_CAPBITS
#define RJ_PT_CAPBITS 9
#endif
constexpr int PT_CAPBITS = RJ_PT_CAPBITS;
constexpr int PT_MAXF    = 1 << PT_CAPBITS;
constexpr int PT_MAXBITS = 9;                     // what the bit plans use per pass (512 digits: runs of 32 tuples per tile)
static_assert(PT_CAPBITS >= PT_MAXBITS && PT_CAPBITS <= 10, "fan-out capacity");
constexpr int PT_FINEBITS = 15;                   // fine (two-digit) histogram: 2^15 bins = 128 KiB of LDS
constexpr int LDS_BYTES  = 160 * 1024;            // per CU (and the most one workgroup may declare)
static_assert(PT_THREADS >= PT_MAXF, "thread d scans digit d");
// static LDS of the partition kernels: staging tile of 8-byte pairs + three digit arrays + wave sums
static_assert(PT_TILE * 8 + 3 * PT_MAXF * 4 + (PT_THREADS / 64) * 4 <= LDS_BYTES,
              "scatter tile does not fit the 160 KiB of LDS");
static_assert((4 << PT_FINEBITS) <= LDS_BYTES, "fine histogram does not fit the LDS");
static_assert(PT_TILE <= 65536, "ranks are packed into 16 bits");
static_assert(PT_ITEMS % 4 == 0 && PT_ITEMS <= 32, "full tiles are loaded as 16-byte vectors; one mask bit per item");

struct PassParams {
    const uint32_t* seg_off;    // [nseg+1] input segments (previous pass' partitions);
                                // nullptr = one segment [0, n)
    const uint32_t* grp_start;  // [nseg+1] exclusive scan of groups per segment (nseg > 1)
    uint32_t        nseg;
    uint32_t        n;          // tuples (single-segment case)
    uint32_t        shift;      // digit = (word0 >> shift) & (F-1)
    uint32_t        fanout_log2;
    uint32_t        tiles_per_group;
    uint32_t*       hist;       // [nseg*F << xcd_log2]  global bin totals
    uint32_t*       cursor;     // [nseg*F << xcd_log2]  write cursors (start = exclusive scan of hist)
    // 3: every partition's range is cut into 8 sub-ranges, one per XCD (workgroup g counts and
    // writes into sub-range g & 7 — workgroups go round-robin over the XCDs), so that the runs
    // next to each other in memory were written through the SAME L2, which can then complete
    // the lines they share before writing them back; 0: one range per partition
    uint32_t        xcd_log2;
    // passes over several segments: workgroup b takes group (b & 7) * ceil(G / 8) + (b >> 3), i.e.
    // every XCD walks its own contiguous eighth of the groups — the workgroups that write into
    // one segment's partitions at the same time then share an L2
    uint32_t        xcd_remap;
    // 12-byte-tuple passes that are not the last one also write the NEXT pass' digit of every
    // tuple as a 16-bit side array (same index as the tuple), so that the next histogram reads
    // 2 bytes per tuple instead of fishing 4-byte keys out of 12-byte tuples; nullptr = none
    uint16_t*       side_out;
    uint32_t        next_shift, next_mask;
    // Input segments that do not lie one behind the other (what a rank holds after the exchange
    // of a sharded join: one run per (source rank, local digit), listed digit-major): segment s
    // is [seg_off[s], seg_end[s]); nullptr = [seg_off[s], seg_off[s + 1])
    const uint32_t* seg_end;
    // ... and several input segments may feed ONE output segment (all runs of a local digit):
    // input segment s belongs to output segment s >> oseg_shift, whose bins / cursors it uses
    uint32_t        oseg_shift;
    // a launch over a sub-range of the segments (a later pass run chunk by chunk): grp_start points at the
    // sub-range's first entry of the table of ALL segments, whose group numbers start at grp_base
    uint32_t        grp_base;
    // Composite digit (stage A of a sharded join: owner rank from the TOP hash bits, first local
    // digit from the LOW ones, one pass for both): hi_shift != 0 =>
    //   digit = ((w >> shift) & ((1 << lo_bits) - 1)) | ((w >> hi_shift) << lo_bits)
    uint32_t        hi_shift, lo_bits;
    // phase cycle counters of the scatter kernels (a diagnostic build, -DRJ_PT_DIAG=1, with
    // RJ_DIAG=3 at run time): thread 0 of every workgroup adds the cycles between consecutive
    // stamps to diag[phase]; nullptr otherwise
    unsigned long long* diag;
};

// digit of hashed key word `w` in this pass (mask = fan-out - 1)
__host__ __device__ inline uint32_t pass_digit(const PassParams& pp, uint32_t w, uint32_t mask) {
    if (pp.hi_shift) return (((w >> pp.shift) & ((1u << pp.lo_bits) - 1u)) | ((w >> pp.hi_shift) << pp.lo_bits)) & mask;
    return (w >> pp.shift) & mask;
}

// ---- Build/probe -------------------------------------------------------------
#ifndef RJ_JN_THREADS
#define RJ_JN_THREADS 512
#endif
#ifndef RJ_JN_SPT
#define RJ_JN_SPT 8
#endif
#ifndef RJ_JN_CAP
#define RJ_JN_CAP 8192
#endif
constexpr int      JN_THREADS = RJ_JN_THREADS;
constexpr int      JN_SPT     = RJ_JN_SPT;             // probe tuples per thread per sub-chunk
constexpr int      JN_SUB     = JN_THREADS * JN_SPT;   // probe tuples per output reservation
constexpr int      JN_CAP     = RJ_JN_CAP;             // LDS table slots (power of two)
constexpr int      JN_RMAX    = JN_CAP / 2;            // build tuples per table (load <= 50 %)
constexpr int      JN_RPT     = (JN_RMAX + JN_THREADS - 1) / JN_THREADS;  // build tuples per thread
// Threads of a join workgroup: tables of 3+ word arrays only fit once per CU, that one workgroup
// then brings all 16 waves itself.
constexpr int jn_threads(int table_words) { return table_words >= 3 ? 2 * JN_THREADS : JN_THREADS; }
// __launch_bounds__ "waves per SIMD" for the join: as many workgroups per CU as the LDS
// table (table_words arrays of JN_CAP words) allows, times threads / 256, capped at 8
constexpr int jn_min_waves(int table_words) {
    int blocks = (160 * 1024) / (JN_CAP * 4 * table_words + JN_CAP + 1024);
    int w = blocks * jn_threads(table_words) / 256;
    return w > 8 ? 8 : (w < 1 ? 1 : w);
}
static_assert(JN_RPT % 4 == 0 && JN_SPT % 4 == 0, "tuples are loaded as 16-byte vectors");
static_assert(JN_CAP * 4 * 4 + JN_CAP + 1024 <= 160 * 1024, "a four-array join table does not fit the LDS");
#ifndef RJ_JN_PPW
#define RJ_JN_PPW 1
#endif
#ifndef RJ_JN_PPW3
#define RJ_JN_PPW3 4
#endif
// Partitions per workgroup: the next partition's loads are issued behind the current one's
// probe (software pipeline).  Tables of two word arrays run two workgroups per CU, whose phases
// overlap by themselves (pipelining them only spilled); tables of 3+ arrays run ONE 1024-thread
// workgroup per CU, which would otherwise serialise load latency, build, probe and emit.
constexpr int jn_ppw(int table_words) { return table_words >= 3 ? RJ_JN_PPW3 : RJ_JN_PPW; }
#ifndef RJ_JN_HEAVY
#define RJ_JN_HEAVY 65536  // (16 K / 32 K / 64 K / 128 K at 1 B rows, Zipf 0.9: join 9.30 / 9.16 / 9.03 / 9.11 ms — profiles/r03_v_*)
#endif
constexpr uint32_t JN_HEAVY   = RJ_JN_HEAVY;           // probe tuples per task before splitting
constexpr uint32_t JN_TARGET_BUILD = JN_RMAX * 3 / 4;  // mean build tuples per final partition

enum StreamMode : int32_t {
    ST_NONE    = 0,
    ST_DENSE32 = 1,
    ST_DENSE64 = 2,
    ST_PAGED32 = 3,  // page images, ROWS32 per page, values from +4
    ST_PAGED64 = 4,  // page images, ROWS64 per page, values from +8
    ST_DENSE96 = 5   // 12-byte records (a three-word wide carry; two-word ones use ST_DENSE64)
};

struct OutStream {
    uint8_t* base;
    int32_t  mode;
    int32_t  pad;
};

struct JoinParams {
    Words           R, S;
    const uint32_t* offR;   // [NP+1]
    const uint32_t* offS;   // [NP+1]
    uint32_t        NP;
    uint32_t        radix_bits;  // low bits of word 0 shared by a partition's tuples
    uint32_t        n_pass;      // radix passes and their bit widths: partition index
    uint32_t        pass_bits[4]; //   q = ((d1 * F2) + d2) * F3 + d3, hash low bits = d1 | d2 << b1 | ...
    OutStream       key, bc, pc; // emitted streams: key, build carry, probe carry
    unsigned long long* out_cursor;
    uint64_t        out_cap;     // rows that fit the stream buffers
    const uint32_t* heavy_tasks; // [n][3] = {partition, s_begin, s_end}
    const uint32_t* n_heavy;
    uint32_t        heavy_grid;  // the first heavy_grid workgroups of the launch take heavy tasks
    int32_t         packR, packS; // R.w[0] / S.w[0] is an array of {hashed key, carry} pairs
    int32_t         aosR, aosS;   // R.w[0] / S.w[0] is an array of 12-byte {hashed key, carry lo, carry hi}
    int32_t         pad;
    unsigned long long* diag;    // phase cycle counters (RJ_DIAG=1 only), else nullptr
};

// ---- wide carries: the records a join emits for one side -> its columns (k_split_records)
struct SplitParams {
    const uint32_t* rec;         // cw words per output row
    uint32_t        cw;
    int32_t         n_cols;
    int32_t         valid_word;  // word of the record that holds the validity bits, -1 = none
    int32_t         pad;
    struct Col {
        uint8_t* out;            // dense values
        uint8_t* valid;          // validity bytes, nullptr = the column has no NULLs
        int32_t  word;           // first word of the column inside the record
        int32_t  width;          // 4 or 8
        int32_t  valid_bit;
        int32_t  paged;          // 1: `out` is a run of Page images (values at their slot), 0: a dense array
    } col[3];
};

// ---- VARCHAR materialisation on the device (rj_varchar_dev.hip) ----------------------------
struct VcRow {
    uint32_t page;  // source page index (first page of a long string's chain)
    uint32_t beg;   // byte offset of the first character inside the page; 0xffffffff = long string
    uint32_t len;   // characters; 0xffffffff = NULL
};
struct VcPage {
    uint32_t first;  // first result row of the page (the row itself for long-string pieces)
    uint32_t nr;     // rows of a normal page
    uint32_t kind;   // 0 = normal page, 1 + k = piece k of a long string
};
constexpr uint32_t VC_CHUNK = 512;  // result rows per fill-rule chunk (one lane walks one chunk)

// ---- Broadcast join (build side fits ONE LDS table): no partitioning at all --------------
struct BcastParams {
    TupleSrc        R, S;        // build / probe tuples straight from the child columns
    OutStream       key, bc, pc; // emitted streams: key, build carry, probe carry
    unsigned long long* out_cursor;
    uint64_t        out_cap;
};

}  // namespace rj
