/*
 * rb3gpu_kernels.h -- hand-written gfx950 (CDNA4, wave64) kernels of the merge engine.
 *
 * Reference functions restated on the device (file:line in the reference tree):
 *   k_tile_hist + k_lf2      rb3_mg_rank_plain prologue, fm-index.c:206-216
 *   k_chain                  rb3_mg_rank1_plain, fm-index.c:160-175 (+ kt_for 217-224)
 *   oct_rank                 rb3_fmi_rank1a -> mr_rank2a -> rope_rank2a -> rle_rank2a
 *                            (fm-index.h:109-112, mrope.c:71-121, rope.c:150-206, rle.c:134-199)
 *   k_pass1 / k_pass2        worker_mgins + rope_insert_run (fm-index.c:237-249, rope.c:114-148)
 *                            and rb3_enc_plain2fmr (fm-index.c:114-137), as one streaming
 *                            interleave that rebuilds the block array
 *   k_export_plain           mr_print_bwt / leaf iteration (mrope.c:133-147, 201-214)
 *   k_ssa_walk/link/final    rb3_ssa_gen (ssa.c:17-81): sampled suffix array of the index
 *
 * No MFMA anywhere: this is integer pointer chasing bound by HBM/L2 latency and bandwidth.
 */
#ifndef RB3GPU_KERNELS_H
#define RB3GPU_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rb3gpu_layout.h"

#define RB3_UNSET (-1LL)
/* One 8-byte word per row of the batch, as in the reference's rb[] (fm-index.c:166-168): until the row
 * is visited it holds LF_B2(row) << 3 | B2[row] with bit 63 set; a walker replaces it by the row's merged
 * position (bit 63 clear), so the record lands in the line the walker has just read. */
#define RB3_ROW_LF   0x8000000000000000ull
#define RB3_ROW_NEXT(x) ((int64_t)(((x) & ~RB3_ROW_LF) >> 3))

struct IdxView {
	const uint64_t *grp64;   // rb3_grp_t viewed as 8 x u64
	const uint4 *slot16;     // rb3_slot_t viewed as 8 x uint4
	const uint64_t *gsm;     // word 6 of every directory entry (slot0 | mask << 32) once more, 8 bytes per group: small enough to stay in the
	                         // L2 (1.3 MB for 1.3 G symbols), so the slot of a position can be asked for before its 64-byte entry has come in
	int64_t n;               // number of symbols
	int64_t m;               // number of sentinels (= acc[1])
	int dense;               // 0: mixed slots.  1: every slot is a bit-plane slot (slot index = position >> 8).
	                         // 2: as 1 and the slot headers carry ABSOLUTE counts (RB3_ABS_HEADERS): rank needs no directory
	int abs;                 // 1: the slot headers carry the whole LF base (an index of fewer than 2^32 symbols): a rank needs the directory's slot word (gsm) to find
	                         // the slot, but not the 64-byte entry with the group's counts.  2 (round 6): the headers carry the LOW 32 BITS of the LF base -- what the
	                         // writers have always stored -- and the rest comes from the table `sb` (lf_base): the same two lines per rank at any size.  0: headers
	                         // relative to the group (rb3gpu_tune abs_limit; the layout of 2^32 symbols and more in rounds 3-5)
	const uint64_t *sb;      // abs = 2: sb[s * 8 + c] = the LF base of c at position s << 31, i.e. the counts of directory entry s << 18 (k_sb_table): a few hundred bytes
};

/* the LF base from a slot header.  wrap = 0: T + hdr (T = 0: the header is the base; T = the group's count: the header is relative to the group).  wrap = 1
 * (IdxView.abs = 2): the header holds the low 32 bits of the base and T is the base 2^31 symbols or less further down -- a base grows by at most one per symbol,
 * so the two differ by less than 2^32 and the difference of their low halves is the difference */
#define RB3_SB_BITS 31
__device__ __forceinline__ uint64_t lf_base(uint64_t T, uint32_t hdr, uint32_t wrap)
{
	return wrap ? T + (uint64_t)(uint32_t)(hdr - (uint32_t)T) : T + (uint64_t)hdr;
}

/* the table of IdxView.sb from the directory of an index (one thread per entry word) */
__global__ void __launch_bounds__(64) k_sb_table(const uint64_t *grp64, int64_t nsb, uint64_t *sb)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nsb * 8) sb[t] = (t & 7) < 6 ? grp64[((t >> 3) << (RB3_SB_BITS - RB3_GRP_BITS)) * 8 + (t & 7)] : 0ull;
}

/* A block array that holds fewer than 2^32 symbols is written with hdr[1..6] = C[a] + #{i < slot start : B[i] = a}, the
 * full LF base, instead of the count relative to the group start: where all slots are bit planes rank(c, k) is then ONE
 * memory request (the slot), which is worth ~16 % of k_chain on such an index; on a run-coded index the 64-byte directory
 * entry with the group's counts is not read any more (the slot is found through the compact copy of the slot words, 8 bytes
 * per group), and the walk of a pangenome build is bound by exactly that: random lines per second behind the L2 (one more
 * random 64-byte line per step costs 21 %, measured; round 1 on the dense index: 19 %).  The group directory is written
 * as always.  Both the builder (device, from the scan totals) and the host (view_of) apply this rule.
 * (nslots, nwin: no longer part of the rule -- rounds 1-3 had absolute headers in bit-plane-only indexes.) */
#define RB3_ABS_LIMIT (1LL << 32)
#define RB3_ABS_HEADERS(ntot, lim) ((ntot) < (lim))

struct Acc7 { int64_t a[7]; };

/* ----------------------------------------------------------------------------------------- */
/* cross-lane helpers for an OCTET (8 consecutive lanes of a wave64), DPP only, no LDS         */
/* ----------------------------------------------------------------------------------------- */

template<int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

/* sum over the 8 lanes of an octet; every lane gets the total */
__device__ __forceinline__ uint32_t oct_sum(uint32_t v)
{
	v += dpp_mov<0x141>(v); // row_half_mirror: lane i <- lane 7-i
	v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
	v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
	return v;
}

/* value of octet lane 0 in every lane of the octet */
__device__ __forceinline__ uint32_t oct_bcast0(uint32_t v, int j)
{
	uint32_t a = dpp_mov<0x00>(v);  // quad_perm [0,0,0,0]
	uint32_t b = dpp_mov<0x114>(a); // row_shr:4
	return (j & 4) ? b : a;
}

/* exclusive prefix sum over the 8 lanes of an octet */
__device__ __forceinline__ uint32_t oct_exscan(uint32_t v, int j)
{
	// an inclusive scan over the ROW of 16 lanes (two octets) with four DPP adds -- lanes shifted in from outside the row add 0 --,
	// then the lower octet's total (lane 7 of the row) comes off the lanes of the upper one: 6 instructions where a step that
	// keeps to its octet (move, select, add) makes 9
	(void)j;
	uint32_t s = v;
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false); // row_shr:1
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false); // row_shr:2
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false); // row_shr:4
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xf, false); // row_shr:8
	s -= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x157, 0xf, 0xc, false); // row_newbcast:7 into lanes 8..15 (banks 2 and 3)
	return s - v;
}

/* the value of lane `l` (0..7) of the octet, l different from octet to octet: the LDS crossbar (ds_bpermute), not a vector instruction */
__device__ __forceinline__ uint32_t oct_pick(uint32_t v, int l)
{
	const int lane = (int)(threadIdx.x & 63u);
	return (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~7) + l) << 2, (int)v);
}

/* the same three for a group of LPW lanes per query: an octet (8) or a QUAD (4 lanes, 16 queries per wave: k_chain's
 * instruction stream is shared by twice as many walkers) */
template<int LPW> __device__ __forceinline__ uint32_t grp_sum(uint32_t v)
{
	if (LPW == 8) return oct_sum(v);
	v += dpp_mov<0xB1>(v); // quad_perm [1,0,3,2]
	v += dpp_mov<0x4E>(v); // quad_perm [2,3,0,1]
	return v;
}

template<int LPW> __device__ __forceinline__ uint32_t grp_bcast0(uint32_t v, int j)
{
	if (LPW == 8) return oct_bcast0(v, j);
	return dpp_mov<0x00>(v); // quad_perm [0,0,0,0]
}

template<int LPW> __device__ __forceinline__ uint32_t grp_exscan(uint32_t v, int j)
{
	if (LPW == 8) return oct_exscan(v, j);
	uint32_t inc = v, t;
	t = dpp_mov<0x111>(inc); if (j >= 1) inc += t; // row_shr:1
	t = dpp_mov<0x112>(inc); if (j >= 2) inc += t; // row_shr:2
	return inc - v;
}

/* inclusive prefix sum over the 64 lanes of a wave with DPP adds (6 instructions, no LDS crossbar round trips; the
 * shuffle formulation costs ~45 instructions and six dependent ds_bpermute latencies) */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1 (lanes shifted in from outside the row of 16 add 0)
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
	return v;
}

/* true in every active lane?  (the lane mask of the comparison against EXEC: __all() goes through an integer per lane -- two more vector instructions) */
__device__ __forceinline__ bool wave_all(bool p)
{
	return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true);
}

/* the value of one given lane (the same for the whole wave) through a scalar register */
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int l)
{
	return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}

/* the value of the lane below (lane 0 gets 0): DPP wave_shr:1 */
__device__ __forceinline__ uint32_t wave_up1(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}

/* waves that work independently inside a block (rebuild, settle): LDS traffic between its lanes only needs program order
 * (the LDS executes a wave's instructions in order), not a workgroup barrier -- and no wait for global
 * loads in flight, which is what makes the loads of the next stage overlap */
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* the value of the lane above (lane 63 gets 0): DPP wave_shl:1 */
__device__ __forceinline__ uint32_t wave_down1(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

/* inclusive prefix maximum over the 64 lanes (unsigned values; same DPP ladder as wave_incl_scan) */
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
	uint32_t t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = v > t ? v : t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = v > t ? v : t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = v > t ? v : t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = v > t ? v : t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;
	t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;
	return v;
}

/* ----------------------------------------------------------------------------------------- */
/* rank: #{i < k : B[i] = c} + C[c], eight lanes per query                                     */
/* ----------------------------------------------------------------------------------------- */

struct RankLoad { // everything that depends on k only, so loads can be issued before c is known
	uint64_t gw;     // word j of the group entry
	uint64_t sm;     // slot0 | mask << 32 of the group entry
	uint4 sl;        // slice j of the slot
	uint32_t koff;   // k & 8191
};

/* round trip 1: the group directory entry (one 64-B line, two requests) */
__device__ __forceinline__ void oct_rank_issue_grp(const IdxView &ix, int64_t k, int j, RankLoad &r)
{
	const int64_t g = k >> RB3_GRP_BITS;
	r.koff = (uint32_t)k & (RB3_GRP - 1);
	r.gw = ix.abs == 2 ? ix.sb[(k >> RB3_SB_BITS) * 8 + j] : ix.grp64[g * 8 + j];
	r.sm = ix.grp64[g * 8 + 6];
}

/* round trip 2: the slot (one 128-B line, 16 B per lane) */
__device__ __forceinline__ void oct_rank_issue_slot(const IdxView &ix, int j, RankLoad &r)
{
	const uint32_t lw = r.koff >> RB3_WIN_BITS;
	const uint32_t slot0 = (uint32_t)r.sm, mask = (uint32_t)(r.sm >> 32);
	const uint32_t s = slot0 + __popc(mask & ((2u << lw) - 1u)) - 1u;
	r.sl = ix.slot16[(int64_t)s * 8 + j];
}

__device__ __forceinline__ void oct_rank_issue(const IdxView &ix, int64_t k, int j, RankLoad &r)
{
	oct_rank_issue_grp(ix, k, j, r);
	oct_rank_issue_slot(ix, j, r);
}

/* Counting in a run slot with packed 16-bit arithmetic (v_pk_*).  A run code is (end - 1) << 3 | sym with `end` the offset, from the
 * slot start, just past the run (rb3gpu_layout.h: CUMULATIVE ends since round 6; rounds 1-5 stored lengths).  The run of a code
 * therefore covers [end of the code before it, its own end): a lane gets the starts of its six runs from its own three dwords and
 * ONE cross-lane read (the last end of the lane below; 0 in lane 0) where lengths needed their sum, a six-step exclusive scan over the
 * octet and a running base -- about 25 vector instructions and the scan's dependent DPP chain less per rank (the walkers' step is bound by
 * instruction issue and by the length of its own dependent chain).  Unused codes (sym 7) sit behind the last used one and repeat its
 * end: empty runs whose symbol never equals c.  Offsets and ends are <= 8192: every quantity fits 16 bits unsigned.
 * cnt = #{i < off : sym_i == c} for this lane's share of the slot: sum over its runs of c of max(min(off, end) - start, 0). */
typedef short rb3_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short rb3_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rb3_s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(rb3_s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(rb3_s16x2 v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ uint32_t pk_sum16(uint32_t v, uint32_t add) // add + low half + high half (v_dot2_u32_u16)
{
	return __builtin_amdgcn_udot2(__builtin_bit_cast(rb3_u16x2, v), __builtin_bit_cast(rb3_u16x2, 0x00010001u), add, false);
}

/* packed unsigned 16-bit helpers of the run-slot decode:
 * pk_subsat: a > b ? a - b : 0 per half (v_pk_sub_u16 clamp -- subtraction and the clamp at 0 in one instruction);
 * pk_dot: add + a.lo * b.lo + a.hi * b.hi (v_dot2_u32_u16 -- with b in {0, 1} per half: mask and sum in one instruction) */
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(rb3_u16x2, a), __builtin_bit_cast(rb3_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(rb3_u16x2, a), __builtin_bit_cast(rb3_u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_dot(uint32_t a, uint32_t b, uint32_t add)
{
	return __builtin_amdgcn_udot2(__builtin_bit_cast(rb3_u16x2, a), __builtin_bit_cast(rb3_u16x2, b), add, false);
}
/* the exclusive ends of the two codes of a dword: (code >> 3) + 1 per half */
__device__ __forceinline__ uint32_t pk_ends(uint32_t w)
{
	return as_u32(__builtin_bit_cast(rb3_s16x2, __builtin_bit_cast(rb3_u16x2, w) >> 3) + as_s16x2(0x00010001u));
}
/* the high half of v in the lane below of the octet, in the HIGH half of the result; 0 in octet lane 0 (DPP row_shr:1 hands lane 8 of a
 * row the value of lane 7 -- the other octet --, hence the select) */
__device__ __forceinline__ uint32_t oct_prev_end(uint32_t v, int j)
{
	const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); // row_shr:1, lanes shifted in from outside the row: 0
	return j ? t : 0u;
}
/* starts of the two runs of a dword whose ends are `e`, given the ends `below` of the dword before it (only its high half counts) */
__device__ __forceinline__ uint32_t pk_starts(uint32_t e, uint32_t below)
{
	return __builtin_amdgcn_alignbit(e, below, 16); // (e.lo << 16) | below.hi
}

/* a dword of two run codes with its unused ones (sym 7) set to "ends at `total`" (writers: the codes behind the last run repeat its end) */
__device__ __forceinline__ uint32_t run_fill_unused(uint32_t w, uint32_t total)
{
	const uint32_t u = RB3_RUN_CODE(total, 7u);
	if ((w & 7u) == 7u) w = (w & 0xFFFF0000u) | u;
	if ((w >> 16 & 7u) == 7u) w = (w & 0x0000FFFFu) | u << 16;
	return w;
}

/* cnt_a / cnt_b: the runs of c below off_a / off_b (TWO).  MATCH: *match_a = 1 in the lane that holds the symbol AT off_a if that symbol
 * is c, else 0 -- the count at off_a + 1 minus the count at off_a. */
template<bool TWO, bool MATCH, int LPW = 8>
__device__ __forceinline__ void slice_count_pk(const uint4 &sl, const uint4 &sl2, int off_a, int off_b, int c, int j, uint32_t *cnt_a, uint32_t *cnt_b, uint32_t *match_a)
{
	static_assert(LPW == 8, "an octet per slot (the quad variant of rounds 2-5 was slower in every regime and went with the length codes)");
	(void)sl2;
	const uint32_t w[3] = { sl.y, sl.z, sl.w };
	uint32_t e[3], st[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) e[k] = pk_ends(w[k]);
	st[0] = pk_starts(e[0], oct_prev_end(e[2], j)), st[1] = pk_starts(e[1], e[0]), st[2] = pk_starts(e[2], e[1]);
	const uint32_t csplat = (uint32_t)c * 0x00010001u;
	// (offsets are unsigned here: 0 <= off <= the slot's symbols <= 8192)
	const uint32_t ua = (uint32_t)off_a * 0x00010001u, ub = (uint32_t)off_b * 0x00010001u, um = ua + 0x00010001u;
	uint32_t acc_a = 0, acc_b = 0, acc_m = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		// 1 in the halves whose symbol is c: x in 0..7 per half, 1 -sat- x is 1 iff x == 0
		const uint32_t eq1 = pk_subsat(0x00010001u, (w[k] & 0x00070007u) ^ csplat);
		acc_a = pk_dot(pk_subsat(pk_minu(ua, e[k]), st[k]), eq1, acc_a);
		if (MATCH) acc_m = pk_dot(pk_subsat(pk_minu(um, e[k]), st[k]), eq1, acc_m);
		if (TWO) acc_b = pk_dot(pk_subsat(pk_minu(ub, e[k]), st[k]), eq1, acc_b);
	}
	*cnt_a = acc_a;
	if (TWO) *cnt_b = acc_b;
	if (MATCH) *match_a = acc_m - acc_a;
}

/* The two ends of an interval that lie in two (neighbouring) run slots, through ONE instruction stream: off_a is counted in
 * slice sa, off_b in slice sb.  An interval of up to 255 rows against slots of 512+ symbols straddles a slot boundary in one
 * step out of five or ten, and with eight walkers per wave some walker does in a third of the wave's iterations; giving that
 * walker two single decodes made the whole wave run ~300 more instructions.  cnt = #{i < off : sym_i == c}, this lane's share. */
template<int LPW = 8>
__device__ __forceinline__ void slice_count_pk2(const uint4 &sa, const uint4 &sa2, const uint4 &sb, const uint4 &sb2, int off_a, int off_b, int c, int j, uint32_t *cnt_a, uint32_t *cnt_b)
{
	static_assert(LPW == 8, "an octet per slot");
	(void)sa2, (void)sb2;
	const uint32_t wa[3] = { sa.y, sa.z, sa.w }, wb[3] = { sb.y, sb.z, sb.w };
	uint32_t ea[3], eb[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) ea[k] = pk_ends(wa[k]), eb[k] = pk_ends(wb[k]);
	// (the two "ends of the lane below" travel in one DPP read: the high halves of ea[2] and eb[2])
	const uint32_t pv = oct_prev_end(__builtin_amdgcn_perm(eb[2], ea[2], 0x07060302u), j); // hi(ea[2]) | hi(eb[2]) << 16
	const uint32_t sta[3] = { pk_starts(ea[0], pv << 16), pk_starts(ea[1], ea[0]), pk_starts(ea[2], ea[1]) };
	const uint32_t stb[3] = { pk_starts(eb[0], pv), pk_starts(eb[1], eb[0]), pk_starts(eb[2], eb[1]) };
	const uint32_t csplat = (uint32_t)c * 0x00010001u;
	const uint32_t ua = (uint32_t)off_a * 0x00010001u, ub = (uint32_t)off_b * 0x00010001u; // (unsigned offsets: see slice_count_pk)
	uint32_t acc_a = 0, acc_b = 0;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const uint32_t eqa = pk_subsat(0x00010001u, (wa[k] & 0x00070007u) ^ csplat), eqb = pk_subsat(0x00010001u, (wb[k] & 0x00070007u) ^ csplat);
		acc_a = pk_dot(pk_subsat(pk_minu(ua, ea[k]), sta[k]), eqa, acc_a);
		acc_b = pk_dot(pk_subsat(pk_minu(ub, eb[k]), stb[k]), eqb, acc_b);
	}
	*cnt_a = acc_a, *cnt_b = acc_b;
}

/* number of symbols equal to c among the first `off` symbols of the slot, this lane's share;
 * bit 20 of the result is set in the one lane that holds the symbol AT offset `off` if that symbol is c */
#define RB3_MATCH_BIT 0x100000u
/* bit planes of 32 symbols: #{i < t : sym_i == c} | RB3_MATCH_BIT if 0 <= t < 32 and the symbol at t is c */
__device__ __forceinline__ uint32_t plane_count(const uint4 &sl, int t, int c)
{
	const uint32_t m0 = (c & 1) ? sl.y : ~sl.y, m1 = (c & 2) ? sl.z : ~sl.z, m2 = (c & 4) ? sl.w : ~sl.w;
	const uint32_t m = m0 & m1 & m2;
	const uint32_t at = (t >= 0 && t < 32) ? ((m >> t) & 1u) : 0u;
	t = t < 0 ? 0 : t > 32 ? 32 : t;
	const uint32_t lim = t >= 32 ? 0xFFFFFFFFu : ((1u << t) - 1u);
	return __popc(m & lim) | (at ? RB3_MATCH_BIT : 0u);
}

template<int LPW = 8, bool MATCH = true>
__device__ __forceinline__ uint32_t slice_count(const uint4 &sl, const uint4 &sl2, uint32_t hdr0, uint32_t off, int c, int j)
{
	uint32_t cnt;
	if (!(hdr0 & RB3_SLOT_RLE)) { // bit planes: an octet lane has symbols [32j, 32j+32), a quad lane [64j, 64j+64)
		if (LPW == 8) cnt = plane_count(sl, (int)off - 32 * j, c);
		else cnt = plane_count(sl, (int)off - 64 * j, c) + plane_count(sl2, (int)off - 64 * j - 32, c); // (at most one of the two sets the match bit)
	} else { // run codes, two at a time
		uint32_t cb, mt;
		slice_count_pk<false, MATCH, LPW>(sl, sl2, (int)off, (int)off, c, j, &cnt, &cb, &mt);
		if (MATCH) cnt |= mt << 20; // (RB3_MATCH_BIT: 0 or 1 from at most one lane of the octet)
	}
	return cnt;
}

__device__ __forceinline__ int64_t oct_rank_finish(const RankLoad &r, int c, int j, int abs_hdr)
{
	const uint32_t hdr0 = oct_bcast0(r.sl.x, j);
	const uint32_t off = r.koff - (hdr0 & 0xFFFFu);
	uint32_t part = slice_count<8>(r.sl, r.sl, hdr0, off, c, j) & (RB3_MATCH_BIT - 1u);
	if (abs_hdr == 1) return (int64_t)oct_sum(j == c + 1 ? r.sl.x : 0u) + (int64_t)oct_sum(part); // hdr[c+1] is the whole LF base
	if (abs_hdr == 2) { // ... its low half; the table word of c came in as gw of lane c
		const uint32_t tl = oct_sum(j == c ? (uint32_t)r.gw : 0u), th = oct_sum(j == c ? (uint32_t)(r.gw >> 32) : 0u);
		return (int64_t)(lf_base((uint64_t)th << 32 | tl, oct_sum(j == c + 1 ? r.sl.x : 0u), 1u) + (uint64_t)oct_sum(part));
	}
	if (j == c + 1) part += r.sl.x; // hdr[c+1] = count of c between group start and slot start
	const uint32_t sum = oct_sum(part);
	const uint32_t lo = oct_sum(j == c ? (uint32_t)r.gw : 0u);
	const uint32_t hi = oct_sum(j == c ? (uint32_t)(r.gw >> 32) : 0u);
	return (int64_t)(((uint64_t)hi << 32 | lo) + sum);
}

/* ----------------------------------------------------------------------------------------- */
/* batched rank (tests): one octet per query, all six symbols                                  */
/* ----------------------------------------------------------------------------------------- */

__global__ void __launch_bounds__(256) k_rank_batch(IdxView ix, Acc7 acc, int64_t nq, const int64_t *k, int64_t *ok)
{
	const int lane = threadIdx.x & 63, j = lane & 7;
	int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 3;
	for (; q < nq; q += stride) {
		int64_t kk = k[q];
		kk = kk < 0 ? 0 : kk > ix.n ? ix.n : kk;
		RankLoad r;
		oct_rank_issue(ix, kk, j, r);
		for (int c = 0; c < 6; ++c) {
			int64_t v = oct_rank_finish(r, c, j, ix.abs) - acc.a[c];
			if (j == 0) ok[q * 6 + c] = v;
		}
	}
}

/* ----------------------------------------------------------------------------------------- */
/* LF array of the partial BWT B2 (fm-index.c:206-216)                                         */
/* ----------------------------------------------------------------------------------------- */

/* everything a merge wants cleared before its first kernel, in ONE launch (nine hipMemsetAsync calls cost a launch each: ~45 us per
 * merge of 1.8 ms): up to 8 regions of 16-byte units, each filled with one 32-bit value */
struct FillJobs { void *p[8]; unsigned long long n16[8]; uint32_t val[8]; int n; };

__global__ void __launch_bounds__(256) k_fill_regions(FillJobs jb)
{
	unsigned long long tot = 0;
	for (int i = 0; i < jb.n; ++i) tot += jb.n16[i];
	for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < tot; u += (unsigned long long)gridDim.x * blockDim.x) {
		unsigned long long v = u;
		int i = 0;
		while (v >= jb.n16[i]) v -= jb.n16[i], ++i;
		((uint4*)jb.p[i])[v] = make_uint4(jb.val[i], jb.val[i], jb.val[i], jb.val[i]);
	}
}

/* the compact copy of the directory's slot words (IdxView.gsm), made after every rebuild */
__global__ void __launch_bounds__(256) k_grp_compact(const uint64_t *grp64, int64_t ngrp, uint64_t *gsm, const unsigned long long *skip)
{
	if (skip && (skip[0] | skip[1] | skip[2])) return; // (single-sync merge whose rank phase did not validate: nothing was built)
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g < ngrp) gsm[g] = grp64[g * 8 + 6];
}

#define RB3_TILE 4096 // bytes of B2 per workgroup (256 threads x 16 B)

/* per-tile symbol histogram: tcnt[tile*8 + a], a = 0..5; [6] = #bytes outside 0..5 */
__global__ void __launch_bounds__(256) k_tile_hist(const uint8_t *b2, int64_t n2, uint32_t *tcnt)
{
	__shared__ uint32_t sh[8];
	const int64_t tile = blockIdx.x;
	if (threadIdx.x < 8) sh[threadIdx.x] = 0;
	__syncthreads();
	const int64_t base = tile * RB3_TILE + (int64_t)threadIdx.x * 16;
	uint32_t c[7] = {0, 0, 0, 0, 0, 0, 0};
	if (base + 16 <= n2) {
		const uint4 v = *(const uint4*)(b2 + base);
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t a = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
#pragma unroll
			for (int s = 0; s < 6; ++s) c[s] += (a == (uint32_t)s);
			c[6] += (a > 5u);
		}
	} else {
		for (int i = 0; i < 16 && base + i < n2; ++i) {
			const uint32_t a = b2[base + i];
#pragma unroll
			for (int s = 0; s < 6; ++s) c[s] += (a == (uint32_t)s);
			c[6] += (a > 5u);
		}
	}
#pragma unroll
	for (int s = 0; s < 7; ++s) {
		uint32_t v = c[s];
		for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[s], v);
	}
	__syncthreads();
	if (threadIdx.x < 8) tcnt[tile * 8 + threadIdx.x] = threadIdx.x < 7 ? sh[threadIdx.x] : 0;
}

/* Exclusive scan of records of 8 x u32 (7 used) into records of 8 x u64, three kernels. */
#define RB3_SCAN_CHUNK 1024 // records per workgroup

__global__ void __launch_bounds__(256) k_scan_chunk_totals(const uint32_t *in, int64_t nrec, uint64_t *ctot)
{
	__shared__ uint64_t sh[7];
	if (threadIdx.x < 7) sh[threadIdx.x] = 0;
	__syncthreads();
	const int64_t r0 = (int64_t)blockIdx.x * RB3_SCAN_CHUNK;
	uint64_t s[7] = {0, 0, 0, 0, 0, 0, 0};
	for (int i = threadIdx.x; i < RB3_SCAN_CHUNK; i += 256) {
		const int64_t r = r0 + i;
		if (r < nrec) {
			const uint4 a = *(const uint4*)(in + r * 8), b = *(const uint4*)(in + r * 8 + 4);
			s[0] += a.x, s[1] += a.y, s[2] += a.z, s[3] += a.w, s[4] += b.x, s[5] += b.y, s[6] += b.z;
		}
	}
#pragma unroll
	for (int q = 0; q < 7; ++q) {
		uint64_t v = s[q];
		for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
		if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&sh[q], (unsigned long long)v);
	}
	__syncthreads();
	if (threadIdx.x < 8) ctot[(int64_t)blockIdx.x * 8 + threadIdx.x] = threadIdx.x < 7 ? sh[threadIdx.x] : 0;
}

/* single workgroup: ctot[] -> exclusive prefix in place; total[8] receives the grand totals */
__global__ void __launch_bounds__(256) k_scan_chunks(uint64_t *ctot, int64_t nchunk, uint64_t *total)
{
	__shared__ uint64_t sh[256][7];
	const int t = threadIdx.x;
	const int64_t per = (nchunk + 255) / 256, c0 = (int64_t)t * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
	uint64_t s[7] = {0, 0, 0, 0, 0, 0, 0};
	for (int64_t c = c0; c < c1; ++c)
		for (int q = 0; q < 7; ++q) s[q] += ctot[c * 8 + q];
	for (int q = 0; q < 7; ++q) sh[t][q] = s[q];
	__syncthreads();
	if (t < 7) { // sequential exclusive scan over 256 partials, one thread per column
		uint64_t run = 0;
		for (int i = 0; i < 256; ++i) { uint64_t v = sh[i][t]; sh[i][t] = run; run += v; }
		total[t] = run;
	}
	if (t == 7) total[7] = 0;
	__syncthreads();
	for (int q = 0; q < 7; ++q) s[q] = sh[t][q];
	for (int64_t c = c0; c < c1; ++c)
		for (int q = 0; q < 7; ++q) { uint64_t v = ctot[c * 8 + q]; ctot[c * 8 + q] = s[q]; s[q] += v; }
}

__global__ void __launch_bounds__(256) k_scan_records(const uint32_t *in, int64_t nrec, const uint64_t *ctot, uint64_t *out)
{
	__shared__ uint64_t sh[256][7];
	const int t = threadIdx.x;
	const int64_t r0 = (int64_t)blockIdx.x * RB3_SCAN_CHUNK + (int64_t)t * 4;
	uint32_t v[4][7];
	uint64_t s[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int64_t r = r0 + i;
		if (r < nrec) {
			const uint4 a = *(const uint4*)(in + r * 8), b = *(const uint4*)(in + r * 8 + 4);
			v[i][0] = a.x, v[i][1] = a.y, v[i][2] = a.z, v[i][3] = a.w, v[i][4] = b.x, v[i][5] = b.y, v[i][6] = b.z;
		} else {
#pragma unroll
			for (int q = 0; q < 7; ++q) v[i][q] = 0;
		}
#pragma unroll
		for (int q = 0; q < 7; ++q) s[q] += v[i][q];
	}
#pragma unroll
	for (int q = 0; q < 7; ++q) sh[t][q] = s[q];
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) { // Hillis-Steele inclusive scan over threads
		uint64_t add[7];
#pragma unroll
		for (int q = 0; q < 7; ++q) add[q] = t >= d ? sh[t - d][q] : 0;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 7; ++q) sh[t][q] += add[q];
		__syncthreads();
	}
	uint64_t run[7];
#pragma unroll
	for (int q = 0; q < 7; ++q) run[q] = ctot[(int64_t)blockIdx.x * 8 + q] + sh[t][q] - s[q];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int64_t r = r0 + i;
		if (r < nrec) {
#pragma unroll
			for (int q = 0; q < 7; ++q) { out[r * 8 + q] = run[q]; run[q] += v[i][q]; }
			out[r * 8 + 7] = 0;
		}
	}
}

/* lf2[i] = (C2[a] + #{i' < i : B2[i'] = a}) << 3 | a, a = B2[i]   (fm-index.c:211-216) */
__global__ void __launch_bounds__(256) k_lf2(const uint8_t *b2, int64_t n2, const uint64_t *tpre, const uint64_t *tot, uint64_t *lf2)
{
	__shared__ uint64_t shbase[6];
	__shared__ uint64_t shw[4][2];
	const int64_t tile = blockIdx.x;
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	if (t < 6) { // C2[t] = number of symbols smaller than t in the whole batch (totals of the scan)
		uint64_t c2 = 0;
		for (int a = 0; a < t; ++a) c2 += tot[a];
		shbase[t] = c2 + tpre[tile * 8 + t];
	}
	const int64_t base = tile * RB3_TILE + (int64_t)t * 16;
	uint8_t sym[16];
	int nv = 0;
	if (base + 16 <= n2) {
		const uint4 v = *(const uint4*)(b2 + base);
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int i = 0; i < 16; ++i) sym[i] = (w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
		nv = 16;
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i) { sym[i] = base + i < n2 ? b2[base + i] : 7; nv += (base + i < n2); }
	}
	// per-thread counts packed as 3 x 16-bit fields in two words (a = 0,1,2 | 3,4,5)
	uint64_t c0 = 0, c1 = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t a = sym[i];
		if (a < 3) c0 += 1ull << (16 * a); else if (a < 6) c1 += 1ull << (16 * (a - 3));
	}
	// exclusive scan over the 256 threads of the tile
	uint64_t i0 = c0, i1 = c1;
	for (int d = 1; d < 64; d <<= 1) {
		uint64_t u0 = __shfl_up(i0, d), u1 = __shfl_up(i1, d);
		if (lane >= d) i0 += u0, i1 += u1;
	}
	if (lane == 63) shw[wv][0] = i0, shw[wv][1] = i1;
	__syncthreads();
	uint64_t p0 = i0 - c0, p1 = i1 - c1;
	for (int w = 0; w < wv; ++w) p0 += shw[w][0], p1 += shw[w][1];
	// walk the 16 symbols
	uint64_t out[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t a = sym[i] < 6 ? sym[i] : 0;
		const uint64_t word = a < 3 ? p0 : p1;
		const uint32_t sh = 16 * (a < 3 ? a : a - 3);
		const uint64_t in_tile = (word >> sh) & 0xFFFFull;
		out[i] = RB3_ROW_LF | (shbase[a] + in_tile) << 3 | a;
		if (a < 3) p0 += 1ull << sh; else p1 += 1ull << sh;
	}
	if (nv == 16) {
#pragma unroll
		for (int i = 0; i < 16; i += 2) {
			ulonglong2 v; v.x = out[i], v.y = out[i + 1];
			*(ulonglong2*)(lf2 + base + i) = v;
		}
	} else {
		for (int i = 0; i < nv; ++i) lf2[base + i] = out[i];
	}
}

/* ----------------------------------------------------------------------------------------- */
/* LF chains (fm-index.c:160-175), one octet per walker                                        */
/* ----------------------------------------------------------------------------------------- */

/* A WALKER follows one string of the batch right to left: row kb of B2 and its insertion point
 * ka in B1 advance together, ka' = C1[c] + rank_B1(c, ka), kb' = LF_B2(kb), and pos[kb] = ka+kb
 * is recorded (fm-index.c:166-173).  The sentinel rows 0..m2-1 start walkers with the exact
 * ka = m1 (fm-index.c:164).  To get parallelism out of long strings, extra walkers start in the
 * middle of strings knowing only lo = 0 <= ka <= hi = n1.  Both bounds obey the same recurrence
 * and LF is monotone, so lo <= ka <= hi stays true; once lo == hi (the suffix read so far no
 * longer occurs in B1's text) the walker is exact and starts recording.  At the end of its own
 * segment (the start row of the next walker to the left) an inexact walker stops; an exact one
 * keeps going through rows nobody has recorded yet (the unresolved head of the next segment)
 * until it meets a recorded row or the start of the string.  Every recorded value is exact, so
 * two walkers that overlap write identical numbers and no ordering between workgroups is needed.
 *
 * Where the extra walkers start:
 *   LIST = false  rows kb >= m2 with kb % 2^logM == 0 (no knowledge of the text needed: this is
 *                 what the reference's signature rb3_fmi_merge_plain(r, len, bwt) allows)
 *   LIST = true   an explicit list {row, ka0, nsteps, flags} supplied by the caller, normally the
 *                 rows of text positions spaced M apart (sampled inverse suffix array, which the
 *                 host suffix sorter has at hand); nsteps = length of the walker's own segment.
 *                 RB3_WK_CHECK starts it in "check every row" mode (fix-up walkers).
 *                 stop_row >= 0 (multi-GPU text-range sharding): the rows from stop_row on belong to
 *                 another GPU; any walker reaching it stops there and, if exact, leaves the value it
 *                 arrived with in *arrive for the neighbour rank.
 */
struct Walker { int64_t row, ka0, nsteps, flags; };
#ifdef RB3_PROF_WAVES
#define RB3_PROF_WAVES_MAX 16384
__device__ unsigned long long g_prof_wave[4 * RB3_PROF_WAVES_MAX];
__device__ unsigned long long g_prof_wave_n;
#endif
#define RB3_MISC_BADW 40 /* word of the merge's counter block (rb3gpu.hip: misc[]; k_chain's nsteps is misc + 1) that counts list entries which are not walkers of the batch */
#define RB3_WK_CHECK 2

/* row words are read and written with agent-scope relaxed atomics: a record must reach the L2 where walkers
 * on other compute units (and XCDs) look for it, and must not be torn */
__device__ __forceinline__ int64_t ld_pos(const int64_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_pos(int64_t *p, int64_t v)
{
	// written through (sc1: the other XCDs' walkers must find it in memory) and streaming (nt: the line does not stay in this XCD's L2 -- a record
	// is looked at once or never, and the 70 MB of records of a merge pushed the index's slots out of the L2s: k_chain 0.62 -> 0.58-0.60 ms per launch)
#ifdef RB3_EXP_NO_RECNT /* kernel experiment: the store of rounds 1-4 */
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
	asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
#endif
}

/* rank with the symbol known before the loads are issued: every lane of the octet fetches the one
 * directory word cnt[c] (same address: one request) instead of reducing it across lanes */
struct RankLoadC {
	uint64_t gc;     // grp.cnt[c]
	uint64_t sm;     // slot0 | mask << 32
	uint4 sl;        // slice j of the slot (quads: slice 2j)
	uint4 sl2;       // quads: slice 2j + 1
	uint32_t koff;   // k & 8191
	uint32_t sidx;   // index of the slot (mixed indexes)
	uint32_t wrap;   // gc is a word of IdxView.sb and the headers hold low halves (lf_base)
};

template<int LPW>
__device__ __forceinline__ void octc_load_slot(const IdxView &ix, int64_t s, int j, RankLoadC &r)
{
	if (LPW == 8) r.sl = ix.slot16[s * 8 + j];
	else r.sl = ix.slot16[s * 8 + 2 * j], r.sl2 = ix.slot16[s * 8 + 2 * j + 1];
}

template<bool DENSE, int LPW = 8>
__device__ __forceinline__ void octc_issue_grp(const IdxView &ix, int64_t k, int c, int j, RankLoadC &r)
{
	const int64_t g = k >> RB3_GRP_BITS;
	r.koff = (uint32_t)k & (RB3_GRP - 1);
	r.wrap = ix.abs == 2 ? 1u : 0u, r.gc = 0;
	if (ix.abs == 2) r.gc = ix.sb[(k >> RB3_SB_BITS) * 8 + c];   // (a table of a few hundred bytes)
	if (DENSE) octc_load_slot<LPW>(ix, k >> RB3_WIN_BITS, j, r); // every window is its own slot and carries the LF base: no directory lookup
	else if (ix.abs) r.sm = ix.gsm[g];                            // the headers carry the LF base: only the slot word, from its compact copy
	else r.gc = ix.grp64[g * 8 + c], r.sm = ix.grp64[g * 8 + 6];
}

template<bool DENSE, int LPW = 8>
__device__ __forceinline__ void octc_issue_slot(const IdxView &ix, int j, RankLoadC &r)
{
	if (DENSE) return;
	const uint32_t lw = r.koff >> RB3_WIN_BITS;
	const uint32_t s = (uint32_t)r.sm + __popc((uint32_t)(r.sm >> 32) & ((2u << lw) - 1u)) - 1u;
	r.sidx = s;
	octc_load_slot<LPW>(ix, (int64_t)s, j, r);
}

/* the upper bound of an interval whose lower bound is being fetched as `lo`: the two usually lie in the same slot
 * (an interval of <= 255 rows against slots of >= 512 symbols), and then there is nothing to fetch */
template<bool DENSE, int LPW = 8>
__device__ __forceinline__ void octc_issue_slot_hi(const IdxView &ix, int j, RankLoadC &r, const RankLoadC &lo)
{
	if (DENSE) return;
	const uint32_t lw = r.koff >> RB3_WIN_BITS;
	const uint32_t s = (uint32_t)r.sm + __popc((uint32_t)(r.sm >> 32) & ((2u << lw) - 1u)) - 1u;
	r.sidx = s;
	if (s != lo.sidx) octc_load_slot<LPW>(ix, (int64_t)s, j, r);
	else { r.sl = lo.sl; if (LPW == 4) r.sl2 = lo.sl2; }
}

/* header word c + 1 of the slot (the count of c before the slot), in the lane that holds it: octet lane j has word j, quad lane j words 2j and 2j+1 */
template<int LPW>
__device__ __forceinline__ uint32_t octc_hdr_c(const RankLoadC &r, int c, int j)
{
	if (LPW == 8) return j == c + 1 ? r.sl.x : 0u;
	return 2 * j == c + 1 ? r.sl.x : 2 * j + 1 == c + 1 ? r.sl2.x : 0u;
}

/* the count of symbol c in the slot's header (word 0 of slice c + 1), in every lane of the group */
template<int LPW>
__device__ __forceinline__ uint32_t octc_hdr_pick(const RankLoadC &r, int c, int j)
{
	if (LPW == 8) return oct_pick(r.sl.x, c + 1);
	return grp_sum<LPW>(octc_hdr_c<LPW>(r, c, j));
}

/* both ends of an interval that lies inside ONE run slot, from one decode */
template<int LPW = 8>
__device__ __forceinline__ void octc_finish_pair(const RankLoadC &rl, uint32_t koff_hi, uint32_t hdr0, int c, int j, int64_t *lo_n, int64_t *hi_n)
{
	const int base = (int)(hdr0 & 0xFFFFu);
	uint32_t ca, cb;
	uint32_t mt;
	slice_count_pk<true, false, LPW>(rl.sl, rl.sl2, (int)rl.koff - base, (int)koff_hi - base, c, j, &ca, &cb, &mt);
	uint32_t v = ca | cb << 16; // both fit 16 bits (counts inside a slot of at most 8192 symbols)
	v = grp_sum<LPW>(v);
	const uint64_t hb = lf_base(rl.gc, octc_hdr_pick<LPW>(rl, c, j), rl.wrap); // (the header may be the whole LF base: 32 bits)
	*lo_n = (int64_t)(hb + (v & 0xFFFFu)), *hi_n = (int64_t)(hb + (v >> 16));
}

/* the same with the slot's offset in its group given (derived from the directory's mask: no header word needed) */
template<int LPW = 8>
__device__ __forceinline__ void octc_finish_pair_at(const RankLoadC &rl, int off_lo, int off_hi, int c, int j, int64_t *lo_n, int64_t *hi_n, bool have_hdr = false, uint32_t hdr_c = 0u)
{
	uint32_t ca, cb, mt;
	const uint64_t hb = lf_base(rl.gc, have_hdr ? hdr_c : octc_hdr_pick<LPW>(rl, c, j), rl.wrap); // (the header may be the whole LF base: 32 bits; asked for first: it crosses the lanes while the codes are counted)
	slice_count_pk<true, false, LPW>(rl.sl, rl.sl2, off_lo, off_hi, c, j, &ca, &cb, &mt);
	uint32_t v = ca | cb << 16;
	v = grp_sum<LPW>(v);
	*lo_n = (int64_t)(hb + (v & 0xFFFFu)), *hi_n = (int64_t)(hb + (v >> 16));
}

/* LF(c, k) for the group's query; *match = 1 iff the symbol at offset k itself is c (then the suffix
 * at row k extends by c: used to advance an interval [k, k+1) with a single rank) */
template<bool DENSE, int LPW = 8, bool MATCH = true>
__device__ __forceinline__ int64_t octc_finish(const RankLoadC &r, int c, int j, uint32_t *match)
{
	uint32_t part;
	if (DENSE) { // bit planes, slot start = window start, absolute header (RB3_ABS_HEADERS)
		const int off = (int)(r.koff & (RB3_WIN - 1));
		if (LPW == 8) part = plane_count(r.sl, off - 32 * j, c);
		else part = plane_count(r.sl, off - 64 * j, c) + plane_count(r.sl2, off - 64 * j - 32, c);
		const uint32_t base = grp_sum<LPW>(octc_hdr_c<LPW>(r, c, j)), sum = grp_sum<LPW>(part);
		*match = sum >> 20;
		return (int64_t)lf_base(r.gc, base, r.wrap) + (int64_t)(sum & (RB3_MATCH_BIT - 1u));
	} else {
		const uint32_t hdr0 = grp_bcast0<LPW>(r.sl.x, j);
		part = slice_count<LPW, MATCH>(r.sl, r.sl2, hdr0, r.koff - (hdr0 & 0xFFFFu), c, j);
	}
	const uint32_t sum = grp_sum<LPW>(part), base = grp_sum<LPW>(octc_hdr_c<LPW>(r, c, j)); // (the header may be the whole LF base: 32 bits)
	*match = sum >> 20;
	return (int64_t)(lf_base(r.gc, base, r.wrap) + (sum & (RB3_MATCH_BIT - 1u)));
}

/* Tentative records (TENT = true).  An inexact walker whose interval [lo, hi) has shrunk to a few
 * rows -- k = hi - lo suffixes of B1's text start with what it has read -- knows ka up to a small
 * number: ka = lo + d, where d counts how many of those k suffixes are smaller than the new one.
 * The k suffixes and the new one are extended by the same symbols as the walk goes on, so their
 * relative order never changes: d stays put while all k survive, and when some of them drop out
 * because B1 has another symbol there, d loses the dropped ones that were below it.  Such a walker
 * therefore records RB3_TENT | sid << 38 | (lo + kb), where sid names a STRETCH of rows that share
 * one unknown d, opens a new stretch at every drop-out (an EVENT record: previous stretch, lo, k, c --
 * three stores; WHICH rows dropped is worked out afterwards by k_events), and goes on.  k = 1 is the
 * common case for a genome merged into an index holding one close relative and has a fast path with
 * ONE rank per step (hi' = lo' + [B1[lo] == c]); k > 1 (several close relatives indexed) costs the two
 * ranks a wide walker pays anyway.  Whoever later walks into those rows knowing more settles d instead
 * of redoing the rows:
 *   an exact walker      -> del of that stretch = 1 + (its value - the recorded lo + kb)  and stops;
 *   a tentative walker   -> a LINK record: its own stretch, offset (both intervals contain ka, so the
 *                           two unknowns differ by the difference of the two lo)          and stops;
 * a tentative walker that meets a final value learns its own d the same way.  k_resolve follows the
 * event/link chains, k_pos_finalize_check[_rows] rewrites the tentative records.  The critical path
 * of a merge drops from the longest variant-free stretch of the batch to about one segment.
 *
 * Concurrency.  Records become visible late (lane-parked, written through), so a follower only a
 * few rows behind a tentative walker would read "unvisited", record its own value and never notice
 * the tags.  Three measures: (1) a walker records tentatively only once it is RB3_TENT_MIN_AGE
 * steps old, so whoever follows it into its segment is that many rows behind -- provided the
 * follower's own segment was at least that long, which the host guarantees for the walker lists it
 * generates (LIST) but not for the automatic split (segments are geometric there); (2) without that
 * guarantee every record is an unsigned 64-bit atomic MIN, and the encoding orders
 * final < tentative < unvisited, so the better-informed value survives whatever the arrival order
 * (atomics sustain ~25 G/s on this chip against ~70 G/s for plain stores, hence only there);
 * (3) the host counts unsettled tentative records after the launch and, if there are any, redoes
 * the rank phase without tentative records (rb3gpu.hip).  Correctness therefore never depends on
 * timing.
 */
#define RB3_TENT      (1LL << 62)
#define RB3_TENT_PBITS 38             /* bits of a merged position in a tentative record: merges up to 2^38 symbols */
#define RB3_TENT_MASK ((1LL << RB3_TENT_PBITS) - 1)
#define RB3_TENT_IDS  (1 << 24)       /* stretch ids per merge */
#define RB3_TENT_POISON (RB3_TENT_IDS - 1) /* the id of records whose stretch could not be allocated: never settled */
#define RB3_TENT_KMAX 255             /* widest interval that is tracked tentatively with the masks inside the stretch records (256 bits) */
#define RB3_TENT_QMAX 8               /* ... and with masks of 256 Q bits in an array of their own, Q = 2, 4, 8 (more than 255 relatives: below) */
#define RB3_TENT_KMAX_TOP (256 * RB3_TENT_QMAX - 1)
#ifndef RB3_TENT_MIN_AGE
#define RB3_TENT_MIN_AGE 16u      /* walker lists (segments of a guaranteed length).  32 in rounds 2-4; round 5, with the records written as a stream: 16 walks
                                     6 % fewer steps (the pre-roll of every walker), one merge of 38 k redone on a crowded GPU against 12 of 38 k at 32; at 12 and 8
                                     the walkers reach their segments on intervals too wide to record and the walk gets SLOWER (profiles/r5_ab_min_age.txt) */
#endif
#ifndef RB3_TENT_MIN_AGE_AUTO
#define RB3_TENT_MIN_AGE_AUTO 64u /* automatic split: the segments are geometric, and with 32 one merge in a hundred of the
                                     soak left records unsettled and was redone */
#endif
/* One 64-byte record per stretch, so that following a dependency path costs one memory round trip per
 * stretch (k_resolve).  w0, w1: how the unknown of the stretch follows from another one.
 *   w0 = type << 62 | previous stretch << 38 | lo (EVENT only)
 *   EVENT: some rows of the previous stretch's interval [lo, lo + kk) do not hold c and dropped out;
 *          w1 = kk | c << 16 | tp << 20 (tp: the text position of the row at which the walker noted it, TEXT walkers; the
 *          stretch begins at text position tp - 1).  k_events turns that into the 256-bit mask of the dropped
 *          rows (the walker itself only notes the event: three stores, no loads), and
 *          d = d(prev) - #{dropped rows with index < d(prev)}
 *   LINK:  d = d(prev) + (int32)w1
 * del: 0 = unknown, else 1 + d as settled by a walker; child: 1 + the stretch that depends on this one
 * (there is at most one), 0 = none. */
typedef struct {
	uint64_t w0, w1;
	int32_t del, child;
	uint32_t mask[8];
	uint32_t pad[2];
} rb3_stretch_t; /* 64 bytes; the first 24 must be zero before a merge */
#define RB3_EV_TP_SHIFT 20
#define RB3_DEP_EVENT 1ull
#define RB3_DEP_LINK  2ull
#define RB3_DEP_W0(type, prev, lo) ((uint64_t)(type) << 62 | (uint64_t)(uint32_t)(prev) << RB3_TENT_PBITS | (uint64_t)(lo))
#define RB3_DEP_PREV(w0) ((int)((w0) >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1))
#define RB3_TENT_BLOCK 8             /* walkers with several matching suffixes get their stretch ids in aligned blocks of
                                        this many, from the lower half of the table (counter sidctr[0]); walkers with a
                                        unique match never see a drop-out and take single ids from the upper half
                                        (counter sidctr[1]) */
#define RB3_TENT_HALF (RB3_TENT_IDS / 2)
#ifndef RB3_TENT_CHUNK
#define RB3_TENT_CHUNK 8             /* ids (a whole number of blocks) a walker takes from a counter at a time */
#endif
/* Where the blocks come from.  One counter for all walkers means that every atomicAdd of a launch goes to the same word, and at 150
 * relatives one wave iteration in seven needs a new block: k_chain took 163 ms per 152-genome build with blocks of 8 from one counter,
 * 140 with 32 ids per atomic, 129 with 128 -- but every id handed out and not used is cleared and looked at by the settle kernels
 * (rank phase 193 / 181 / 219 ms).  So: RB3_TENT_NCTR counters, each on a cache line of its own (mctr[c * 32]), a wave uses the one
 * of its number, and chunk n of counter c is the ids [(n * NCTR + c) * CHUNK, + CHUNK): the counters advance at the same rate, so the
 * ids in use stay dense, and k_tent_extent turns the largest of them into the extent the settle kernels scan (sidctr[0]). */
#ifndef RB3_TENT_NCTR
#define RB3_TENT_NCTR 64
#endif
__device__ __forceinline__ uint32_t tent_take_chunk(uint32_t *sidctr, uint32_t *mctr, uint32_t c)
{
	if (mctr == nullptr) return atomicAdd(sidctr, (uint32_t)RB3_TENT_CHUNK);
	const uint32_t n = atomicAdd(&mctr[c * 32u], 1u);
	return n >= (uint32_t)(RB3_TENT_HALF / (RB3_TENT_NCTR * RB3_TENT_CHUNK)) ? 0xFFFFFF00u : (n * (uint32_t)RB3_TENT_NCTR + c) * (uint32_t)RB3_TENT_CHUNK; // (a full table: an id beyond every limit)
}

__global__ void k_tent_extent(const uint32_t *mctr, uint32_t *sidctr)
{
	uint32_t v = threadIdx.x < RB3_TENT_NCTR ? mctr[threadIdx.x * 32u] : 0u;
	for (int o = 32; o; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64); v = v > t ? v : t; }
	if (threadIdx.x == 0) {
		const unsigned long long e = (unsigned long long)v * RB3_TENT_NCTR * RB3_TENT_CHUNK;
		sidctr[0] = e < (unsigned long long)RB3_TENT_HALF ? (uint32_t)e : (uint32_t)RB3_TENT_HALF;
	}
}

template<bool TENT> __device__ __forceinline__ void rec_pos(int64_t *p, int64_t v, bool vis)
{
	if (TENT) atomicMin((unsigned long long*)p, (unsigned long long)v); // final < tentative < unvisited (bit 63)
	else if (vis) st_pos(p, v);  // written through: other walkers may be waiting to see it
#ifdef RB3_EXP_PLAINNT /* kernel experiment: ... as a stream as well */
	else __builtin_nontemporal_store(v, p);
#else
	else *p = v;                 // one walker per string, nobody ever looks: let the L2 keep the line it just read
#endif
}

/* Rows of [lo, lo + kk) that do NOT hold c, from the slice of a slot this lane already has in registers:
 * OR them into the octet's 256-bit mask D[8] (LDS; bit i <=> row lo + i).  Index i sits at slot offset
 * off0 + i (off0 may be negative: the rows before the slot are somebody else's).  Called for the slot of
 * lo and for the slot of hi, which together hold all of [lo, hi) when hi - lo < 256. */
template<int NW = 8>
__device__ __forceinline__ void drops_from_slot(const uint4 &sl, uint32_t hdr0, int off0, int kk, int c, int j, uint32_t *D)
{
	if (!(hdr0 & RB3_SLOT_RLE)) { // bit planes: this lane holds slot offsets [32j, 32j + 32)
		const uint32_t m0 = (c & 1) ? sl.y : ~sl.y, m1 = (c & 2) ? sl.z : ~sl.z, m2 = (c & 4) ? sl.w : ~sl.w;
		uint32_t nm = ~(m0 & m1 & m2);
		const int ib = 32 * j - off0; // index of bit 0 of this word
		const int t0 = ib < 0 ? -ib : 0, t1 = kk - ib < 32 ? kk - ib : 32;
		if (t0 < t1) {
			nm &= (t1 >= 32 ? 0xFFFFFFFFu : (1u << t1) - 1u) & ~((1u << t0) - 1u);
			if (nm) {
				const int wi = ib >> 5, sh = ib & 31; // arithmetic shift: wi may be -1
				if (wi >= 0 && wi < NW && (nm << sh)) atomicOr(&D[wi], nm << sh);
				if (sh && wi + 1 >= 0 && wi + 1 < NW && (nm >> (32 - sh))) atomicOr(&D[wi + 1], nm >> (32 - sh));
			}
		}
	} else { // six run codes per lane; run i covers [the end of the code before it, its own end) -- unused codes repeat the last end: empty
		int pos = (int)(oct_prev_end(pk_ends(sl.w), j) >> 16);
#pragma unroll
		for (int i = 0; i < 6; ++i) {
			const uint32_t word = i < 2 ? sl.y : i < 4 ? sl.z : sl.w, e = (i & 1) ? word >> 16 : word & 0xFFFFu;
			const int end = (int)(e >> 3) + 1;
			if ((int)(e & 7u) != c && (e & 7u) != 7u && end > pos) {
				int a = pos - off0, b = end - off0; // index range of this run
				a = a < 0 ? 0 : a, b = b > kk ? kk : b;
				if (a < b) { // (nearly always one row of one word: a relative that dropped out)
					int w = a >> 5;
					const int wl = (b - 1) >> 5, x1l = ((b - 1) & 31) + 1;
					if (w < NW) atomicOr(&D[w], (w == wl && x1l < 32 ? (1u << x1l) - 1u : 0xFFFFFFFFu) & ~((1u << (a & 31)) - 1u));
#pragma unroll 1
					for (++w; w <= wl && w < NW; ++w) atomicOr(&D[w], w == wl && x1l < 32 ? (1u << x1l) - 1u : 0xFFFFFFFFu);
				}
			}
			pos = end;
		}
	}
}

/* mask[j] of every EVENT stretch: the rows of [lo, lo + kk) that do not hold c (one octet per stretch).
 * The two slots that hold the rows are those of lo and of lo + kk. */
/* mctr != NULL: the extent of the ids in use is still in the 64 counters the walkers took their chunks from (see tent_take_chunk): every
 * block works it out for itself and block 0 leaves it in sidctr_out[0] for the kernels behind this one and for the host (k_tent_extent's
 * job, without its launch) */
/* U stretches per octet and iteration (their loads asked for together).  Measured in round 5 (profiles/r5_ab_settle_widths.txt): U = 2 changes
 * nothing, and neither do wider launches of this or the other settle kernels -- the pass is not waiting for memory: its counters say 18 M
 * vector instructions per launch (the run decode of drops_from_slot for every event), i.e. ~30 of its ~40 us are the vector units. */
#ifndef RB3_EV_UNROLL
#define RB3_EV_UNROLL 1
#endif
__global__ void __launch_bounds__(256) k_events(IdxView ix, rb3_stretch_t *tab, const uint32_t *sidctr, const uint32_t *mctr = nullptr, uint32_t *sidctr_out = nullptr)
{
	constexpr int U = RB3_EV_UNROLL;
	__shared__ uint32_t dmask[U][32 * 8];
	const int j = threadIdx.x & 7;
	uint32_t ext = 0;
	if (mctr != nullptr) {
		uint32_t v = (threadIdx.x & 63) < RB3_TENT_NCTR ? mctr[(threadIdx.x & 63) * 32u] : 0u;
		for (int o = 32; o; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64); v = v > t ? v : t; }
		const unsigned long long e = (unsigned long long)v * RB3_TENT_NCTR * RB3_TENT_CHUNK;
		ext = e < (unsigned long long)RB3_TENT_HALF ? (uint32_t)e : (uint32_t)RB3_TENT_HALF;
		if (blockIdx.x == 0 && threadIdx.x == 0) sidctr_out[0] = ext;
	} else ext = *sidctr;
	const int64_t n = ext < (uint32_t)RB3_TENT_HALF ? ext : RB3_TENT_HALF; // events only happen to ids from blocks
	const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 3;
	for (int64_t sid0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; sid0 < n; sid0 += stride * U) {
		uint64_t w0[U], w1[U];
		bool ev[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t sid = sid0 + u * stride;
			ev[u] = sid < n;
			w0[u] = ev[u] ? tab[sid].w0 : 0ull, w1[u] = ev[u] ? tab[sid].w1 : 0ull;
		}
		bool any = false;
#pragma unroll
		for (int u = 0; u < U; ++u) ev[u] = w0[u] >> 62 == RB3_DEP_EVENT, any |= ev[u];
		if (!any) continue;
		// the slots of lo and of lo + kk: both directory words first, then the slot(s) -- the interval usually lies in ONE slot,
		// which is then read and searched once (the two dependent round trips of a rank, not four)
		int64_t lo[U], k1[U];
		uint64_t sm0[U], sm1[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			lo[u] = (int64_t)(w0[u] & (uint64_t)RB3_TENT_MASK);
			const int kk = (int)(w1[u] & 0xFFFF);
			k1[u] = lo[u] + kk <= ix.n ? lo[u] + kk : lo[u];
			sm0[u] = sm1[u] = 0;
			if (ev[u]) sm0[u] = ix.grp64[(lo[u] >> RB3_GRP_BITS) * 8 + 6], sm1[u] = ix.grp64[(k1[u] >> RB3_GRP_BITS) * 8 + 6]; // (measured: the compact copy IdxView.gsm is no faster here, +1 ms per build)
		}
		uint4 sl0[U], sl1[U];
		bool two[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const uint32_t koff0 = (uint32_t)lo[u] & (RB3_GRP - 1), koff1 = (uint32_t)k1[u] & (RB3_GRP - 1);
			const int64_t s0 = (int64_t)((uint32_t)sm0[u] + __popc((uint32_t)(sm0[u] >> 32) & ((2u << (koff0 >> RB3_WIN_BITS)) - 1u)) - 1u);
			const int64_t s1 = (int64_t)((uint32_t)sm1[u] + __popc((uint32_t)(sm1[u] >> 32) & ((2u << (koff1 >> RB3_WIN_BITS)) - 1u)) - 1u);
			two[u] = s1 != s0 || (lo[u] >> RB3_GRP_BITS) != (k1[u] >> RB3_GRP_BITS);
			sl0[u] = make_uint4(0u, 0u, 0u, 0u);
			if (ev[u]) sl0[u] = ix.slot16[s0 * 8 + j];
			sl1[u] = sl0[u];
			if (ev[u] && two[u]) sl1[u] = ix.slot16[s1 * 8 + j];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			if (!ev[u]) continue;
			uint32_t *D = &dmask[u][(threadIdx.x >> 3) * 8];
			const int kk = (int)(w1[u] & 0xFFFF), c = (int)(w1[u] >> 16 & 7);
			const uint32_t koff0 = (uint32_t)lo[u] & (RB3_GRP - 1), koff1 = (uint32_t)k1[u] & (RB3_GRP - 1);
			D[j] = 0u;
			__builtin_amdgcn_wave_barrier();
			{
				const uint32_t hdr0 = oct_bcast0(sl0[u].x, j);
				drops_from_slot(sl0[u], hdr0, (int)koff0 - (int)(hdr0 & 0xFFFFu), kk, c, j, D);
			}
			if (two[u] && k1[u] != lo[u]) {
				const uint32_t hdr0 = oct_bcast0(sl1[u].x, j);
				drops_from_slot(sl1[u], hdr0, (int)koff1 - (int)(hdr0 & 0xFFFFu) - kk, kk, c, j, D);
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			tab[sid0 + u * stride].mask[j] = D[j];
			__builtin_amdgcn_wave_barrier();
		}
	}
}

/* TEXT = true (with LIST): the batch comes as TEXT-ORDER words tw[t] = row of the suffix at text position t << 3 |
 * the symbol before it (0 at the start of a string) -- the inverse suffix array, which a suffix sorter has at
 * hand -- and Walker.row is the text position a walker starts at.  A walker then streams its words (8 bytes
 * per step out of lines it shares with its next steps) instead of fetching a row word from a random row, and
 * row[] only holds records (all "unvisited" to begin with).  Inside its own segment a walker cannot meet a
 * record (others enter a segment through its start row, which is checked when the walker starts), so the
 * record of the next row is only looked up once the walker has left its segment. */
#define RB3_BEYOND (INT64_MAX / 4 * 3) /* remaining > this: the walker has left its own segment */
/* TEXT = 1: every lane of the octet loads the word it needs (one address per octet; with few walkers per compute
 * unit the lines stay in the L1 for the 16 steps they serve).  TEXT = 2: the octet fetches the words of 8 steps with
 * one 64-byte request and hands them out with cross-lane reads (many walkers per compute unit, i.e. one per short
 * string: the L1 cannot hold a line per walker). */
/* LPW: lanes per walker, 8 (an octet) or 4 (a quad: every lane takes two slices of a slot, and the wave's instruction
 * stream -- what a step costs where the index is run-coded -- serves 16 walkers instead of 8) */
/* I32 (the common step only): index and batch are small enough for 32-bit positions -- headers with the LF base (IdxView.abs), fewer than
 * 2^29 rows in the batch -- so that the step's arithmetic is 32-bit and its addresses are a base and a 32-bit offset */
template<bool LIST, bool DENSE, bool TENT, int TEXT, int LPW = 8, bool I32 = false>
#ifndef RB3_CHAIN_WPE
#define RB3_CHAIN_WPE 5 /* waves per SIMD the register allocation must leave room for: 96 registers.  (Round 6: with the event queue the allocator takes 102 when left
                           alone -- four waves per SIMD; held to 96 it spills ONE pair, outside the loops.  6 -- 80 registers -- spills inside the common step: slower, round 5) */
#endif
__global__ void __launch_bounds__(256, RB3_CHAIN_WPE) k_chain(IdxView b1, int64_t *row, int64_t n2, int64_t m2,
		int logM, const Walker *wl, int64_t nwalk_arg, int64_t stop_row, int64_t *arrive, unsigned long long *qhead, unsigned long long *nsteps, int octs,
		rb3_stretch_t *tab, uint32_t *sidctr, uint32_t sid_limit, const uint64_t *tw, const unsigned long long *nwalk_dev = nullptr, int kmax = RB3_TENT_KMAX, uint32_t *mctr = nullptr, int trec_arg = 0, int64_t *jmet = nullptr)
{
	// jmet (TEXT, TENT, LIST): one word per walker, 1 + the text position of the row at which the walker met somebody's record and settled or
	// linked an unknown (0: it never did) -- with the text positions of the events in the stretch records, all the places where what one
	// walker knew was combined with what another had recorded: k_junction_check verifies the LF relation at every one of them
	const int trec = trec_arg & 1;
#ifdef RB3GPU_TEST_HOOKS
	// test hook (rb3gpu_tune "hide_first"): an EXACT walker does not see the tentative records of a walker's FIRST stretch -- what happens for real when it
	// follows a late walker more closely than records become visible: it records over those rows and settles a LATER stretch (see k_cum)
	const bool hide_first = (trec_arg & 2) != 0;
#endif
	// trec (TEXT only): row[] is indexed by TEXT POSITION instead of by row.  The eight records an octet parks are then eight
	// consecutive words -- one 64-byte store where a record per row is eight random 8-byte stores, each costing a 32-byte sector
	// and, in an index that lives in HBM, most of the step (k_chain 18.2 -> 6.4 ms per 302 M steps with the stores removed) -- and
	// the validation pass gathers them into row order through the suffix array (k_pos_finalize_check_rows).
	const uint32_t myctr = (uint32_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (uint32_t)(RB3_TENT_NCTR - 1); // the id counter of this wave
	const int64_t nwalk = nwalk_dev ? (int64_t)*nwalk_dev : nwalk_arg; // (a list made on the device: its length never went to the host)
	static_assert(LIST || !TEXT, "text-order words need a walker list");
	// ids a walker may take from each half of the stretch table (the whole half unless a test narrows it)
	const uint32_t lim_blocks = sid_limit < (uint32_t)RB3_TENT_HALF ? sid_limit : (uint32_t)RB3_TENT_HALF;
	const uint32_t lim_singles = sid_limit < (uint32_t)(RB3_TENT_POISON - RB3_TENT_HALF) ? sid_limit : (uint32_t)(RB3_TENT_POISON - RB3_TENT_HALF);
	static_assert(LPW == 8 || LPW == 4, "an octet or a quad per walker");
	const int lane = threadIdx.x & 63, j = lane & (LPW - 1);
#ifdef RB3_EXP_PRIO /* kernel experiment: the waves of a SIMD at different issue priorities (they run in a convoy: all of them decode at the same time, at a fifth of the speed, then all wait) */
	if (RB3_EXP_PRIO == 1) { // static, by the wave's slot in its SIMD (HW_ID bits 3:0)
		const uint32_t slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);
		if ((slot & 3u) == 0u) __builtin_amdgcn_s_setprio(0); else if ((slot & 3u) == 1u) __builtin_amdgcn_s_setprio(1); else if ((slot & 3u) == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
	} else if (RB3_EXP_PRIO == 3) { // static, by the layer of the block (blocks arrive round-robin over the compute units)
		const uint32_t layer = (blockIdx.x >> 8) & 3u;
		if (layer == 0u) __builtin_amdgcn_s_setprio(3); else if (layer == 1u) __builtin_amdgcn_s_setprio(2); else if (layer == 2u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
	}
#endif
	// With few walkers the kernel is latency-bound and a wave runs every instruction of every octet
	// it hosts: the host may enable only the first `octs` octets of each wave and launch more waves.
	if (lane / LPW >= octs) return; // (octs counts groups of LPW lanes)
	// The EVENT QUEUE of the common step (round 6).  A drop-out event is three stores by one lane of the octet (the stretch record, the walker's first
	// stretch, the child word of the stretch before), in a branch region of their own, two wave iterations in three late in a pangenome build; and what
	// is asked for behind a written-through store waits for its acknowledgement (vmcnt counts in order): the directory words of the NEXT step.  The
	// event is now noted in LDS -- entry (octet, iteration & 7), written by octet lane 0 -- and goes out with the records of its window of eight
	// iterations, every lane one entry, in front of them (a record never names a stretch whose event is not on its way): seven steps in eight issue no
	// store at all, and an event costs three store instructions per wave and window instead of three per octet and event.
#ifdef RB3_NO_EVQ /* kernel experiment: the events written where they happen (rounds 2-5) */
	constexpr bool EVQ = false;
#else
	constexpr bool EVQ = LIST && !DENSE && TENT && TEXT == 1 && LPW == 8;
#endif
	__shared__ uint4 evq_[EVQ ? 512 : 1]; // entry t: evq_[2t] = w0, w1 of the stretch record; evq_[2t + 1].x = the new stretch id (-1: no event), .y = 1 + the walker's first stretch
	if (EVQ) evq_[2 * threadIdx.x + 1].x = 0xFFFFFFFFu;
	auto evq_flush = [&]() { // every lane its own entry: (octet lane >> 3, iteration & 7 == lane & 7)
		if (EVQ) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (entries are written by other lanes of this wave: program order is all the LDS needs)
			const int ns = (int)evq_[2 * threadIdx.x + 1].x;
			if (ns >= 0) {
				const uint32_t first1 = evq_[2 * threadIdx.x + 1].y;
				const uint4 e = evq_[2 * threadIdx.x];
				const int prev = (int)(e.y >> (RB3_TENT_PBITS - 32)) & (RB3_TENT_IDS - 1);
#ifdef RB3_EXP_NOEVST /* kernel experiment (wrong results, right timing): what do the event stores cost the walk?  1 = none of them, 2 = only the 16 bytes of the record */
				if (RB3_EXP_NOEVST == 2) *(uint4*)&tab[ns].w0 = e;
				(void)first1; (void)prev;
#else
				*(uint4*)&tab[ns].w0 = e;
				tab[ns].pad[0] = first1;
				tab[prev].child = ns + 1;
#endif
				evq_[2 * threadIdx.x + 1].x = 0xFFFFFFFFu;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		}
	};
	const int64_t myoct = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * octs + lane / LPW, noct = (int64_t)gridDim.x * (blockDim.x >> 6) * octs;
	bool firstpull = true;
	const int64_t M = logM > 0 ? (1LL << logM) : 0;
	const int64_t first_marked = M ? ((m2 + M - 1) >> logM << logM) : 0;
	// records must become visible to other walkers only if strings are split (logM < 0 with a list: one walker per string)
	const bool vis = (LIST && logM >= 0) || M != 0;
	bool active = false;
	int gap = 0;            // 0: exact (lo == hi), 1: hi == lo + 1, 2: wider
	int sid = -1;           // stretch id of the tentative records being written, -1: none yet, -2: none to be had
	int sid0 = -1;          // the first stretch this walker opened (the unknown all its later stretches follow from)
	int64_t kb = 0, lo = 0, hi = 0, remaining = 0;
	uint64_t x = 0;         // row word of the current row (requested one step ahead)
	int64_t tp = 0;         // TEXT: text position of the current row
	uint64_t x1 = 0;        // TEXT: word of the next row (requested two steps ahead)
	uint64_t rc = ~0ull;    // TEXT: record word of the current row (unvisited unless looked up)
	uint64_t blk8 = 0;      // TEXT: the words this octet fetches during the current window of 8 iterations (lane j: at iteration phase j),
	                        // loaded with one 64-byte request per octet and window, whatever the L1 does with the lines
	uint32_t steps = 0;
	uint32_t wcur = 0;      // the list entry of the current walker (jmet)
	// Records are written through to memory (agent scope) so that walkers on other XCDs can see them.
	// Each octet parks up to 8 records in its lanes (lane it&7 takes iteration it) and the whole wave
	// flushes them with ONE store instruction every 8 iterations.
	int64_t bkb = -1, bval = 0;
	uint32_t it = 0, age = 0;
	// I32, common step: the slot words (gsm) of the group the NEXT insertion point falls into, asked for as soon as the slot of this step has
	// arrived: its header holds the LF base of c at the slot start, and the step adds at most the slot's symbols (<= 8192) to it, so the next
	// point lies in group base >> 13 or in the one behind it -- two adjacent 8-byte words, one 16-byte request that is in flight while the
	// codes are counted.  The step's chain is then slot -> (slot words | decode) -> slot instead of slot -> decode -> slot words -> slot.
	uint32_t pf_g = 0x80000000u; // group of pf_w.x/.y (pf_w.z/.w: the group behind it); 0x80000000: nothing asked for
	uint4 pf_w = make_uint4(0u, 0u, 0u, 0u);
#ifdef RB3_PROF_STEP
	uint64_t prof_t[7] = {0, 0, 0, 0, 0, 0, 0}, prof_last = 0, prof_u[2] = {0, 0};
#endif
#ifdef RB3_PROF_WAVES
	const unsigned long long prof_wt0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef RB3_PROF
	const uint64_t tstart = __builtin_readcyclecounter();
	unsigned long long prof_nonpair = 0; // iterations in which some group of this wave took the two-decode path
#endif
	for (;;) {
		// ---- refill: every octet without a walker pulls the next one from the queue (rare) ----
		if (!active) {
			// the first walker of every octet is the one of its own number (all octets ask at once when the kernel starts, and a list
			// of text-regular walkers has about one per octet: ~23 k atomics on one word in the first microseconds); later ones
			// come from the queue behind those
			int64_t wid;
#ifdef RB3_EXP_STRIDE /* kernel experiment: the eight octets of a wave take walkers from eight distant parts of the list (the text) instead of eight neighbours */
			if (firstpull) wid = (int64_t)(lane / LPW) * (noct / octs) + myoct / octs, firstpull = false;
#else
			if (firstpull) wid = myoct, firstpull = false;
#endif
			else {
				uint32_t w0 = 0, w1 = 0;
				if (j == 0) {
					unsigned long long w = atomicAdd(qhead, 1ull);
					w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
				}
				w0 = grp_bcast0<LPW>(w0, j), w1 = grp_bcast0<LPW>(w1, j);
				wid = noct + (int64_t)((uint64_t)w1 << 32 | w0);
			}
			if (wid >= nwalk) {
				evq_flush();
				if (bkb >= 0) rec_pos<TENT && !LIST>(&row[bkb], bval, vis);
				break;
			}
			wcur = (uint32_t)wid;
			if (TENT && TEXT && LIST && jmet != nullptr && j == 0) jmet[wid] = 0; // (every list entry is taken by exactly one octet: the table needs no clearing)
			int probe = 0; // TEXT: text distance to a row of the right neighbour's segment that tells whether this walker comes too late (below)
			if (LIST) {
				const Walker w = wl[wid];
				kb = w.row, remaining = w.nsteps;
				probe = (int)(w.flags >> 16 & 0xFF);
				if (!TEXT && kb < 0) continue; // an empty slot of a device-made list
				// not a walker of this batch (a caller's list is checked HERE, by the octet that takes the walker, not by a loop on the host
				// that the device waited for: 40 k entries took the host ~40 us per merge with the chip idle): counted, the host returns EINVAL
				if ((uint64_t)w.row >= (uint64_t)n2 || w.nsteps <= 0) {
					if (j == 0 && !(TEXT && w.row < 0)) atomicAdd(nsteps + (RB3_MISC_BADW - 1), 1ull); // (TEXT, row -1: a per-string list whose string count was wrong: the rows stay unset)
					continue;
				}
				if (TEXT) tp = w.row;
				if (w.ka0 >= 0) lo = hi = w.ka0;
				else if (w.ka0 == -2) lo = hi = b1.m; // RB3GPU_KA_SENTINEL: a sentinel row, ka = acc[1] of the index (fm-index.c:164)
				else lo = 0, hi = b1.n;
			} else {
				remaining = INT64_MAX;
				if (wid < m2) kb = wid, lo = hi = b1.m;
				else kb = first_marked + ((wid - m2) << logM), lo = 0, hi = b1.n;
			}
			gap = hi - lo > 1 ? 2 : (int)(hi - lo);
			age = 0, sid = -1;
			if (TEXT) {
				x = tw[tp], x1 = tw[tp > 0 ? tp - 1 : 0];
				if (TEXT == 2) { // the rest of the current window: this walker's first step runs at phase (it + 1) & 7
					const int64_t a = tp - 2 - (int64_t)((j - (int)(it + 1)) & (LPW - 1));
					blk8 = tw[a > 0 ? a : 0];
				}
				kb = (int64_t)(x >> 3);
				rc = (uint64_t)ld_pos(&row[trec ? tp : kb]);
				if (gap && (int64_t)rc >= 0) continue;
				// A walker that starts LATE (its wave was not resident when the kernel began: it starts when the first waves finish, i.e.
				// just when its right neighbour reaches the end of its segment) must not start at all: the neighbour, finding this
				// walker's rows unrecorded, walks them itself, and a late walker recording behind it leaves a stretch that nobody
				// settles (the merge is then redone).  Its start row is the neighbour's -- recorded if the neighbour has passed -- but
				// only visible some iterations later: the row `probe` positions further right has been visible for that long.
				if (gap && probe > 0 && tp + probe < n2) {
					const int64_t tq = tp + probe;
					const int64_t rp = ld_pos(&row[trec ? tq : (int64_t)(tw[tq] >> 3)]);
					if (rp >= 0) continue;
				}
			} else {
				x = (uint64_t)ld_pos(&row[kb]);
				if (gap && (int64_t)x >= 0) continue; // an inexact walker whose start row somebody has already recorded
			}
			active = true;
		}
		// ---- steps: run until some octet of this wave needs a refill.  Two independent dependency
		// chains meet in a step: kb -> row[kb] -> next row, and ka -> directory entry -> slot -> next ka.
		// The next row's word is requested one step ahead, so the symbol c is known before anything is
		// issued and only the ka chain is on the critical path.  The body is branch-free (selects) except
		// for the second bound of wide walkers and the rare stretch events: a lone wave runs at
		// instruction-issue speed, so instruction count is the cost.
		do {
#ifndef RB3_NO_FAST_STEP
#if !defined(RB3_NO_LEAN32) && !defined(RB3_PROF_STEP)
			// ---- the common step on 32-bit state (I32: the headline's kernel; round 6).  The same step as the loop behind it -- read that one first --, made for the
			// LENGTH of a wave's instruction stream: a lone wave of this kernel needs ~10 cycles per instruction (dependent issue, VALU <-> SALU hand-overs, ~25 branch
			// points per step), five waves per SIMD do not hide that (profiles/r6_lone_wave.txt: 3.6 k cycles per step alone, 6.1 k with five), so what a step costs is
			// its instruction count.  Here: the walker's state as 32-bit words (position, WIDTH of the interval instead of its upper end, text position, the two text
			// words) so that the loop's back edge moves 6 registers instead of 18; ONE counter (scalar) for the iteration number, the walker's age, its steps and what
			// is left of its segment -- the per-walker limits are worked out once, on the way in, and the 64-bit state is written back on the way out.
			if constexpr (I32 && LIST && !DENSE && TENT && TEXT == 1 && LPW == 8) {
#define RB3_BAL(x) __builtin_amdgcn_ballot_w64(x)
				const unsigned long long exm = RB3_BAL(true);
				const unsigned long long m_rc = RB3_BAL((int32_t)(rc >> 32) < 0); // (the record word of the row is only looked up outside the segment: it does not change in here)
				// common steps this walker may take: as long as remaining >= 2 (and it has not left its segment: remaining <= RB3_BEYOND)
				uint32_t bud = (uint64_t)(remaining - 2) < (uint64_t)(RB3_BEYOND - 1) ? (remaining - 1 > 0x3FFFFFFFLL ? 0x3FFFFFFFu : (uint32_t)(remaining - 1)) : 0u;
				// ... and, while it is inexact without a stretch, until it is old enough to open one (the general step does that)
				uint32_t agelim = sid == -1 ? (age >= RB3_TENT_MIN_AGE ? 0u : RB3_TENT_MIN_AGE - age) : 0xFFFFFFFFu;
				// (folded into ONE limit: an exact walker stays exact, so only a walker that is inexact on the way in can be the one that has to open a stretch; if it turns exact
				// before that step it merely leaves the loop once too often)
				if ((uint32_t)hi != (uint32_t)lo && agelim < bud) bud = agelim;
				asm volatile("" : "+v"(bud)); // (a plain number from here on: the compiler folded the selects above into the loop's comparison, two instructions a step)
				uint32_t lo32 = (uint32_t)lo, kq = (uint32_t)hi - (uint32_t)lo; // [lo32, lo32 + kq)
				uint32_t xw = (uint32_t)x, x1w = (uint32_t)x1, tp32 = (uint32_t)tp;
				uint32_t its = (uint32_t)__builtin_amdgcn_readfirstlane((int)it), d = 0u; // (the same in every lane: scalar registers)
				const uint32_t nlastg = (uint32_t)b1.n >> RB3_GRP_BITS, nlastw = ((uint32_t)b1.n >> RB3_WIN_BITS & 31u) + 1u;
				const uint32_t pick1 = (uint32_t)(((lane & ~7) + 1) << 2); // ds_bpermute address of header word 1 of this octet's slot
				// the directory words of the first step's group, asked for here; every later step's are asked for by the step before it, from the slot header -- always the right ones (below)
				pf_g = lo32 >> RB3_GRP_BITS;
				pf_w.x = *(const uint32_t*)((const char*)b1.gsm + (((pf_g < (uint32_t)((b1.n >> RB3_GRP_BITS) + 1) ? pf_g : 0u) << 3) + (((uint32_t)j & 3u) << 2)));
				for (;;) {
					const uint32_t cq = xw & 7u;
					const unsigned long long m_simple = RB3_BAL(d < bud) & RB3_BAL(cq != 0u) & m_rc
						& RB3_BAL((lo32 & (RB3_GRP - 1)) + kq <= (uint32_t)RB3_GRP) & RB3_BAL(kq <= (uint32_t)kmax); // (a width that wraps the sum fails the last test)
					if (m_simple != exm) break;
					++its, ++d;
					const int c = (int)cq;
					const uint32_t g32 = lo32 >> RB3_GRP_BITS;
					uint64_t sm;
					{ // the words asked for during the last step are the ones needed: the point lies at most a slot's symbols (<= 8192) behind the header's LF base, i.e. in its group or the next
						const uint32_t v2 = dpp_mov<0xEE>(pf_w.x), v0 = dpp_mov<0x44>(pf_w.x); // quad_perm [2,3,2,3] / [0,1,0,1]
						const uint32_t sel = g32 != pf_g ? v2 : v0;
						sm = (uint64_t)dpp_mov<0x55>(sel) << 32 | dpp_mov<0x00>(sel);
					}
					const uint32_t xn = *(const uint32_t*)((const char*)tw + ((tp32 >= 2u ? tp32 - 2u : 0u) << 3)); // the word after next (c != 0: tp32 >= 1)
					const uint32_t koff = lo32 & (RB3_GRP - 1);
					const uint32_t mask = (uint32_t)(sm >> 32), lw = koff >> RB3_WIN_BITS;
					const uint32_t mlo = mask & ((2u << lw) - 1u);
					const uint32_t sidx = (uint32_t)sm + __popc(mlo) - 1u;
					const uint32_t w0 = 31u - (uint32_t)__builtin_clz(mlo);
					const uint32_t abv = mask & ~((2u << lw) - 1u);
					const uint32_t nxt = abv ? (uint32_t)__builtin_ctz(abv) : 32u;
					const uint32_t glim = g32 == nlastg ? nlastw : 32u;
					const uint32_t wend = nxt < glim ? nxt : glim;
					const bool rle = wend - w0 > 1u;
					const int off_lo = (int)(koff - (w0 << RB3_WIN_BITS)), off_hi = off_lo + (int)kq;
					const bool same = koff + kq <= (wend << RB3_WIN_BITS);
					const unsigned long long m_pair = RB3_BAL(wend - w0 > 1u) & RB3_BAL(koff + kq <= (wend << RB3_WIN_BITS));
					const bool far = !same && kq > 255u;
					RankLoadC rl;
					rl.gc = 0, rl.wrap = 0u, rl.koff = koff, rl.sidx = sidx, rl.sm = sm, rl.sl2 = make_uint4(0u, 0u, 0u, 0u);
					uint4 slb;
					const uint32_t so = sidx * (uint32_t)sizeof(rb3_slot_t) + (uint32_t)j * 16u; // (fewer than 2^24 slots: the byte offset fits 32 bits)
					rl.sl = *(const uint4*)((const char*)b1.slot16 + so);
					if (m_pair != exm) slb = *(const uint4*)((const char*)b1.slot16 + (so + (same ? 0u : (uint32_t)sizeof(rb3_slot_t)))); // (else every interval of the wave lies inside one run slot: nobody looks at a second one)
					const uint32_t kbw = xw >> 3; // the row
					if ((kq == 0u || sid >= 0) && (uint32_t)j == (its & 7u)) {
						const int64_t myval = (int64_t)((uint64_t)lo32 + (uint64_t)kbw);
						bkb = trec ? (int64_t)tp32 : (int64_t)kbw, bval = kq ? (RB3_TENT | ((int64_t)sid << RB3_TENT_PBITS) | myval) : myval;
					}
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					const uint32_t hdr_c = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((cq << 2) + pick1), (int)rl.sl.x); // the LF base of c at the slot start: the next point lies in its group or in the one behind it
					pf_g = hdr_c >> RB3_GRP_BITS;
					pf_w.x = *(const uint32_t*)((const char*)b1.gsm + ((pf_g << 3) + (((uint32_t)j & 3u) << 2))); // (the directory has a spare word behind the last group's)
					if ((its & 7u) == 7u) { // the end of a window of eight iterations: its events, then its records (a scalar test)
						evq_flush();
						if (bkb >= 0) { rec_pos<false>(&row[bkb], bval, vis); bkb = -1; }
					}
					uint32_t lo_n, hi_n;
					if (m_pair == exm) { // everybody inside one run slot
						uint32_t ca, cb, mt;
						slice_count_pk<true, false, 8>(rl.sl, rl.sl2, off_lo, off_hi, c, j, &ca, &cb, &mt);
						const uint32_t v = oct_sum(ca | cb << 16);
						lo_n = hdr_c + (v & 0xFFFFu), hi_n = hdr_c + (v >> 16);
					} else if (wave_all(rle && (oct_bcast0(slb.x, j) & RB3_SLOT_RLE) != 0u && !far)) { // some interval straddles two run slots: everybody through the two-slot decode
						uint32_t ca, cb;
						slice_count_pk2<8>(rl.sl, rl.sl2, slb, slb, off_lo, same ? off_hi : off_hi - (int)((wend - w0) << RB3_WIN_BITS), c, j, &ca, &cb);
						const uint32_t v = oct_sum(ca | cb << 16);
						const uint32_t hh = oct_pick(slb.x, c + 1);
						lo_n = hdr_c + (v & 0xFFFFu), hi_n = hh + (v >> 16);
					} else if (rle && same) {
						uint32_t ca, cb, mt;
						slice_count_pk<true, false, 8>(rl.sl, rl.sl2, off_lo, off_hi, c, j, &ca, &cb, &mt);
						const uint32_t v = oct_sum(ca | cb << 16);
						lo_n = hdr_c + (v & 0xFFFFu), hi_n = hdr_c + (v >> 16);
					} else { // a bit-plane slot somewhere
						uint32_t match = 0, mh;
						const int64_t l64 = octc_finish<false, 8>(rl, c, j, &match);
						int64_t h64 = kq == 1u ? l64 + match : l64;
						if (kq >= 2u) {
							RankLoadC rh;
							octc_issue_grp<false, 8>(b1, (int64_t)lo32 + (int64_t)kq, c, j, rh);
							octc_issue_slot_hi<false, 8>(b1, j, rh, rl);
							h64 = octc_finish<false, 8, false>(rh, c, j, &mh);
						}
						lo_n = (uint32_t)l64, hi_n = (uint32_t)h64;
					}
					const uint32_t kn = hi_n - lo_n;
					if (sid >= 0 && kn - 1u < kq - 1u) { // (1 <= kn < kq, unsigned: false for kq = 0 and 1) // some matching suffixes are not preceded by c: a new stretch (an EVENT, noted in LDS)
						int ns = sid + 1;
						if (sid == RB3_TENT_POISON) ns = RB3_TENT_POISON;
						else if ((ns & (RB3_TENT_CHUNK - 1)) == 0) {
							uint32_t s0 = 0;
							if (j == 0) s0 = tent_take_chunk(sidctr, mctr, myctr);
							s0 = oct_bcast0(s0, j);
							ns = s0 + RB3_TENT_CHUNK <= lim_blocks ? (int)s0 : RB3_TENT_POISON;
						}
						if (j == 0 && ns != RB3_TENT_POISON) {
							const uint64_t ew0 = RB3_DEP_W0(RB3_DEP_EVENT, sid, (uint64_t)lo32), ew1 = (uint64_t)kq | (uint64_t)c << 16 | (uint64_t)tp32 << RB3_EV_TP_SHIFT;
							if (EVQ) {
								const uint32_t qi = (threadIdx.x & ~7u) + (its & 7u);
								evq_[2 * qi] = make_uint4((uint32_t)ew0, (uint32_t)(ew0 >> 32), (uint32_t)ew1, (uint32_t)(ew1 >> 32));
								*(uint2*)&evq_[2 * qi + 1] = make_uint2((uint32_t)ns, (uint32_t)sid0 + 1u);
								__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
							} else {
								tab[ns].w0 = ew0, tab[ns].w1 = ew1;
								tab[ns].pad[0] = (uint32_t)sid0 + 1u;
								tab[sid].child = ns + 1;
							}
						}
						sid = ns;
					}
					tp32 -= 1u, xw = x1w, x1w = xn, lo32 = lo_n, kq = kn;
				}
				// back to the general step's 64-bit state (d == 0: nothing happened)
				it += d, age += d, steps += d, remaining -= (int64_t)d;
				lo = (int64_t)lo32, hi = (int64_t)lo32 + (int64_t)kq, gap = kq > 1u ? 2 : (int)kq;
				x = (uint64_t)xw, x1 = (uint64_t)x1w, tp = (int64_t)tp32, kb = (int64_t)(xw >> 3);
#undef RB3_BAL
			} else
#endif
			if (LIST && !DENSE && TENT && TEXT == 1) for (;;) { // (the common steps of a wave run back to back in this loop of their own: through the do-while's condition the
				// compiler's structured control flow took every one of them round the OUTER loop's header as well -- ~35 instructions and six taken branches per step, round 6)
				// ---- the common step, straight-line.  A genome walked through an index of its relatives spends nine steps in ten in
				// one state: inside its own segment (so nobody has recorded the row and nothing ends here), not at a sentinel, its
				// stretch -- if it records tentatively -- already open.  Everything the general step below tests for on the way
				// (a record met, the end of the segment, a stretch to open, the row after the segment to look up) is then known
				// not to happen, and a wave runs every test whether it fires or not: ~35 branch regions per iteration, a third of
				// the issue slots of a kernel that is bound by instruction issue (5 cycles per instruction at 3-6 waves per SIMD).
				// This body does what the general one does in that state and nothing else; if ANY group of the wave is in another
				// state the whole wave takes the general step, which handles everything.
				const uint32_t cq = (uint32_t)x & 7u;
				const uint64_t kq = I32 ? (uint64_t)((uint32_t)hi - (uint32_t)lo) : (uint64_t)(hi - lo);
				const uint32_t kq32 = kq > 0xFFFFull ? 0xFFFFu : (uint32_t)kq;
				// simple: inside its own segment, not at a sentinel, the row unvisited, no stretch to open (a walker that is inexact, without a stretch, old
				// enough -- and narrow enough, which the last test asks of everybody: a wider interval, a walker in its first dozen steps, may end several
				// slots further on), and the interval inside one GROUP (one that reaches into the next needs a second directory entry: the general step).
				// Every test is a comparison whose lane mask is combined on the scalar unit (a bool per lane that is an AND of comparisons costs two
				// more vector instructions before it can be voted on).
#define RB3_BAL(x) __builtin_amdgcn_ballot_w64(x)
				const unsigned long long exm = RB3_BAL(true);
				const unsigned long long m_simple = RB3_BAL((uint64_t)(remaining - 2) < (uint64_t)(RB3_BEYOND - 1)) & RB3_BAL(cq != 0u) & RB3_BAL((int32_t)(rc >> 32) < 0)
					& ~(RB3_BAL(gap != 0) & RB3_BAL(sid == -1) & RB3_BAL(age >= RB3_TENT_MIN_AGE))
					& RB3_BAL(((uint32_t)lo & (RB3_GRP - 1)) + kq32 <= (uint32_t)RB3_GRP) & RB3_BAL(kq32 <= (uint32_t)kmax);
				if (m_simple != exm) break; // somebody is in another state: one general step for the whole wave (it handles everything), then back here
				{
#ifdef RB3_PROF_STEP /* kernel experiment: where does an iteration of the common step spend its cycles?  (s_memtime at four points) */
					const uint64_t pt0 = __builtin_amdgcn_s_memtime();
#endif
					++it; // (both kinds of step flush the records and events of a window of eight iterations at its end: nothing is left over from a general step)
					const int c = (int)cq;
					// round trip 1: the slot word of lo's group from the compact copy (an L2 hit); the 64-byte entry's count for c is asked
					// for at the same time but only needed at the very end
					const int64_t g = I32 ? (int64_t)((uint32_t)lo >> RB3_GRP_BITS) : lo >> RB3_GRP_BITS;
					RankLoadC rl;
#ifndef RB3_NO_GSM_PF
					const uint32_t pf_d = ((uint32_t)lo >> RB3_GRP_BITS) - pf_g; // 0 or 1 if the words asked for during the last step are the ones needed
#ifndef RB3_NO_TA_DIET
					// (lane j of a quad holds dword j of the 16 bytes asked for: four DPP broadcasts inside the quad put the pair together)
					if (I32 && RB3_BAL(pf_d <= 1u) == exm) {
						const uint32_t v2 = dpp_mov<0xEE>(pf_w.x), v0 = dpp_mov<0x44>(pf_w.x); // quad_perm [2,3,2,3] / [0,1,0,1]: the word of the group behind / of the group itself in lanes 0, 1
						const uint32_t sel = pf_d ? v2 : v0;
						rl.sm = (uint64_t)dpp_mov<0x55>(sel) << 32 | dpp_mov<0x00>(sel);
					}
#else
					if (I32 && RB3_BAL(pf_d <= 1u) == exm) rl.sm = pf_d ? ((uint64_t)pf_w.w << 32 | pf_w.z) : ((uint64_t)pf_w.y << 32 | pf_w.x);
#endif
					else
#endif
					if (I32) rl.sm = *(const uint64_t*)((const char*)b1.gsm + (((uint32_t)lo >> RB3_GRP_BITS) << 3));
					else rl.sm = b1.gsm[g];
					rl.gc = 0, rl.wrap = (!I32 && b1.abs == 2) ? 1u : 0u;
					if (!I32 && !b1.abs) rl.gc = b1.grp64[g * 8 + c]; // (headers relative to the group: rb3gpu_tune abs_limit)
					if (!I32 && b1.abs == 2) rl.gc = b1.sb[(lo >> RB3_SB_BITS) * 8 + c]; // (2^32 symbols or more: the base's upper half from the table)
					const int64_t tpn = I32 ? (int64_t)((uint32_t)tp - 1u) : tp - 1; // (c != 0: there is a symbol before this one, so tp >= 1)
					uint64_t xn;                                      // the word after next
#ifndef RB3_NO_TA_DIET
					// (I32: fewer than 2^29 rows, so a text-order word is its low half -- a dword per lane: the address unit of a compute unit handles a wave's request
					// lane by lane and dword by dword, whatever the lanes share, and it is what this kernel fills: round 6)
					if (I32) xn = (uint64_t)*(const uint32_t*)((const char*)tw + (((uint32_t)tp >= 2u ? (uint32_t)tp - 2u : 0u) << 3));
#else
					if (I32) xn = *(const uint64_t*)((const char*)tw + (((uint32_t)tp >= 2u ? (uint32_t)tp - 2u : 0u) << 3));
#endif // (read as a stream -- nt -- it was 8 % SLOWER: the L1 no longer serves the 15 steps that share a line; one 64-byte request per octet and eight steps, handed out by ds_bpermute as TEXT = 2 does: 33 % slower, profiles/r5_ab_text_words_block.txt)
					else xn = tw[tpn > 0 ? tpn - 1 : 0];
					const int64_t kbn = I32 ? (int64_t)((uint32_t)x1 >> 3) : (int64_t)(x1 >> 3);
					rl.koff = (uint32_t)lo & (RB3_GRP - 1);
					const uint32_t mask = (uint32_t)(rl.sm >> 32), lw = rl.koff >> RB3_WIN_BITS;
					const uint32_t mlo = mask & ((2u << lw) - 1u);    // slot starts at or below lo's window (bit 0 is always set)
					rl.sidx = (uint32_t)rl.sm + __popc(mlo) - 1u;
					// where the slot starts and ends and what kind it is follow from the mask alone (a slot of more than one window is
					// a run slot, a single window is bit planes), and so does whether the upper end of the interval lies in it too or
					// in the slot behind it (same group: see `simple`); an exact walker is the empty interval [lo, lo)
					const uint32_t w0 = 31u - (uint32_t)__builtin_clz(mlo); // (mlo != 0: bit 0)
					// the slot ends where the next one starts -- the lowest mask bit above lo's window --, at the end of the group or, in
					// the last group, with the window of position n
					const uint32_t abv = mask & ~((2u << lw) - 1u);
					const uint32_t nxt = abv ? (uint32_t)__builtin_ctz(abv) : 32u;
					const bool lastg = I32 ? ((uint32_t)lo >> RB3_GRP_BITS) == ((uint32_t)b1.n >> RB3_GRP_BITS) : g == (b1.n >> RB3_GRP_BITS);
					const uint32_t glim = lastg ? (uint32_t)(b1.n >> RB3_WIN_BITS & 31) + 1u : 32u;
					const uint32_t wend = nxt < glim ? nxt : glim;
					const bool rle = wend - w0 > 1u;
					const int off_lo = (int)(rl.koff - (w0 << RB3_WIN_BITS)), off_hi = off_lo + (int)kq32;
					const bool same = rl.koff + kq32 <= (wend << RB3_WIN_BITS);
					const unsigned long long m_pair = RB3_BAL(wend - w0 > 1u) & RB3_BAL(rl.koff + kq32 <= (wend << RB3_WIN_BITS)); // (rle && same: voted on here, where the comparisons are made)
					const bool far = !same && kq32 > 255u; // (wide masks: an interval of more than 255 rows may end beyond the NEXT slot too: the general decode)
					// the slot in which the interval ENDS, asked for at the same time: the next one, or the same one again (a second request for a line
					// that is on its way costs a tag look-up; a load only some lanes make costs a branch region, copies and a wait of its own)
					uint4 slb, slb2 = make_uint4(0u, 0u, 0u, 0u);
					if (LPW == 8) rl.sl2 = make_uint4(0u, 0u, 0u, 0u);
					if (I32) { // (fewer than 2^24 slots: the byte offset fits 32 bits)
						const uint32_t so = rl.sidx * (uint32_t)sizeof(rb3_slot_t) + (uint32_t)j * (LPW == 8 ? 16u : 32u);
						const uint32_t sob = so + (same ? 0u : (uint32_t)sizeof(rb3_slot_t));
#ifdef RB3_EXP_SLOT_NT /* kernel experiment: the slot lines read as a stream (they are 98 MB of random lines against 4 MB of L2 per XCD) */
						{ typedef uint32_t rb3_u32x4 __attribute__((ext_vector_type(4)));
						  const rb3_u32x4 t = __builtin_nontemporal_load((const rb3_u32x4*)((const char*)b1.slot16 + so)); rl.sl = make_uint4(t.x, t.y, t.z, t.w); }
#else
						rl.sl = *(const uint4*)((const char*)b1.slot16 + so);
#endif
#ifndef RB3_NO_SKIP_DUP
						if (m_pair == exm) slb = make_uint4(0u, 0u, 0u, 0u); // every interval of the wave inside one run slot: nobody looks at a second one
						else
#endif
						slb = *(const uint4*)((const char*)b1.slot16 + sob);
						if (LPW == 4) rl.sl2 = *(const uint4*)((const char*)b1.slot16 + (so + 16u));
						if (LPW == 4) slb2 = *(const uint4*)((const char*)b1.slot16 + (sob + 16u));
					} else {
						octc_load_slot<LPW>(b1, (int64_t)rl.sidx, j, rl);
						const int64_t sb = (int64_t)rl.sidx + (same ? 0 : 1);
						if (LPW == 8) slb = b1.slot16[sb * 8 + j];
						else slb = b1.slot16[sb * 8 + 2 * j], slb2 = b1.slot16[sb * 8 + 2 * j + 1];
					}
#ifdef RB3_EXP_PRIO
					if (RB3_EXP_PRIO == 2) { asm volatile("" :: "v"(rl.sl.x), "v"(slb.x)); __builtin_amdgcn_s_setprio(0); } // the slot is asked for: nothing to do but wait
#endif
#ifdef RB3_PROF_STEP
					asm volatile("s_nop 0" :: "v"(rl.sidx));
					const uint64_t pt1 = __builtin_amdgcn_s_memtime(); // the directory word has arrived, the slot is requested
#endif
					++steps;
					const int64_t myval = I32 ? (int64_t)((uint64_t)(uint32_t)lo + (uint64_t)(uint32_t)kb) : lo + kb;
					// (once a stretch is open the interval is at most KMAX wide and the walker old enough, for the rest of its life)
					if ((gap == 0 || sid >= 0) && j == (int)(it & (uint32_t)(LPW - 1)))
						bkb = trec ? tp : kb, bval = gap ? (RB3_TENT | ((int64_t)sid << RB3_TENT_PBITS) | myval) : myval;
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RB3_EXP_PRIO
					if (RB3_EXP_PRIO == 2) __builtin_amdgcn_s_setprio(3); // the slot is here: through the decode and up to the next slot request ahead of the waves that would only wait again
#endif
					uint32_t hdr_c = 0u;
#ifndef RB3_NO_GSM_PF
					if (I32 && LPW == 8) { // (before the record store: what is asked for behind a written-through store waits for its acknowledgement)
						hdr_c = oct_pick(rl.sl.x, c + 1);
						pf_g = hdr_c >> RB3_GRP_BITS;
#ifndef RB3_NO_TA_DIET
						// one dword per lane -- lane j of a quad dword j of the two words -- instead of all 16 bytes in every lane: a quarter of the address unit's work
						pf_w.x = *(const uint32_t*)((const char*)b1.gsm + ((pf_g << 3) + (((uint32_t)j & 3u) << 2))); // (the directory has a spare word behind the last group's)
#else
						const uint2 *pfp = (const uint2*)((const char*)b1.gsm + (pf_g << 3)); // (the directory has a spare word behind the last group's)
						const uint2 pa = pfp[0], pb = pfp[1];
						pf_w = make_uint4(pa.x, pa.y, pb.x, pb.y);
#endif
					}
#endif
					// the records of the last eight steps go out here, where nothing is asked for during the whole decode: the store is slow
					// (written through) and whatever is asked for after it waits for its acknowledgement (vmcnt counts in order)
					if ((it & (uint32_t)(LPW - 1)) == (uint32_t)(LPW - 1)) { // (the events of the window in front of its records)
						evq_flush();
#ifdef RB3_EXP_NORECST /* kernel experiment (wrong results, right timing): the common step without its record stores */
						bkb = -1;
#else
						if (bkb >= 0) { rec_pos<false>(&row[bkb], bval, vis); bkb = -1; }
#endif
					}
#ifdef RB3_PROF_STEP
					asm volatile("s_nop 0" :: "v"(rl.sl.x));
					const uint64_t pt2 = __builtin_amdgcn_s_memtime(); // the slot has arrived
#endif
					int64_t lo_n, hi_n;
#ifdef RB3_PROF_STEP
					if (__ballot(!(rle && same)) != 0ull) prof_t[5] += 1;
#endif
#ifndef RB3_NO_GSM_PF
					if (m_pair == exm) octc_finish_pair_at<LPW>(rl, off_lo, off_hi, c, j, &lo_n, &hi_n, I32 && LPW == 8, hdr_c); // (rle && same, everybody)
#else
					if (m_pair == exm) octc_finish_pair_at<LPW>(rl, off_lo, off_hi, c, j, &lo_n, &hi_n); // (rle && same, everybody)
#endif
					else if (wave_all(rle && (grp_bcast0<LPW>(slb.x, j) & RB3_SLOT_RLE) != 0u && !far)) { // some walker's interval straddles two run slots: everybody through the two-slot decode
						uint32_t ca, cb;
						slice_count_pk2<LPW>(rl.sl, rl.sl2, slb, slb2, off_lo, same ? off_hi : off_hi - (int)((wend - w0) << RB3_WIN_BITS), c, j, &ca, &cb);
						uint32_t v = ca | cb << 16;
						v = grp_sum<LPW>(v);
						RankLoadC rb2;
						rb2.sl = slb, rb2.sl2 = slb2;
						const uint32_t hl = octc_hdr_pick<LPW>(rl, c, j), hh = octc_hdr_pick<LPW>(rb2, c, j); // (headers of 32 bits where they carry the LF base)
						lo_n = (int64_t)(lf_base(rl.gc, hl, rl.wrap) + (v & 0xFFFFu)), hi_n = (int64_t)(lf_base(rl.gc, hh, rl.wrap) + (v >> 16));
					} else if (rle && same) octc_finish_pair_at<LPW>(rl, off_lo, off_hi, c, j, &lo_n, &hi_n);
					else { // a bit-plane slot somewhere
						uint32_t match = 0, mh;
						lo_n = octc_finish<false, LPW>(rl, c, j, &match);
						hi_n = gap == 1 ? lo_n + match : lo_n;
						if (gap == 2) {
							RankLoadC rh;
							octc_issue_grp<false, LPW>(b1, hi, c, j, rh);
							octc_issue_slot_hi<false, LPW>(b1, j, rh, rl);
							hi_n = octc_finish<false, LPW, false>(rh, c, j, &mh);
						}
					}
					const int64_t kn = I32 ? (int64_t)(int32_t)((uint32_t)hi_n - (uint32_t)lo_n) : hi_n - lo_n;
					if (gap == 2 && sid >= 0 && kn >= 1 && kn < (int64_t)kq) { // some matching suffixes are not preceded by c: a new stretch (see below)
						int ns = sid + 1;
						if (sid == RB3_TENT_POISON) ns = RB3_TENT_POISON;
						else if ((ns & (RB3_TENT_CHUNK - 1)) == 0) { // (the next block of the chunk is simply ns)
							// (asking for the next chunk ahead of time, so that nobody waits for the atomic: measured 5 % SLOWER)
							uint32_t s0 = 0;
							if (j == 0) s0 = tent_take_chunk(sidctr, mctr, myctr);
							s0 = grp_bcast0<LPW>(s0, j);
							ns = s0 + RB3_TENT_CHUNK <= lim_blocks ? (int)s0 : RB3_TENT_POISON;
						}
						if (j == 0 && ns != RB3_TENT_POISON) {
							const uint64_t ew0 = RB3_DEP_W0(RB3_DEP_EVENT, sid, lo), ew1 = kq | (uint64_t)c << 16 | (uint64_t)tp << RB3_EV_TP_SHIFT;
							if (EVQ) { // noted in LDS; it goes out with the records of this window (evq_flush)
								const uint32_t qi = (threadIdx.x & ~7u) + (it & 7u);
								evq_[2 * qi] = make_uint4((uint32_t)ew0, (uint32_t)(ew0 >> 32), (uint32_t)ew1, (uint32_t)(ew1 >> 32));
								*(uint2*)&evq_[2 * qi + 1] = make_uint2((uint32_t)ns, (uint32_t)sid0 + 1u);
								__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
							} else {
								tab[ns].w0 = ew0, tab[ns].w1 = ew1;
								tab[ns].pad[0] = (uint32_t)sid0 + 1u;
								tab[sid].child = ns + 1;
							}
						}
						sid = ns;
					}
					++age, --remaining;
					tp = tpn, x = x1, x1 = xn;
					kb = kbn, gap = kn > 1 ? 2 : (int)kn;
					if (I32) lo = (int64_t)(uint32_t)lo_n, hi = (int64_t)(uint32_t)hi_n;
					else lo = lo_n, hi = hi_n;
#ifdef RB3_PROF_STEP
					{
						asm volatile("s_nop 0" :: "v"((uint32_t)lo));
						const uint64_t pt3 = __builtin_amdgcn_s_memtime();
						prof_t[0] += pt1 - pt0, prof_t[1] += pt2 - pt1, prof_t[2] += pt3 - pt2, prof_t[3] += 1;
						if ((it & 7u) == 0u) prof_u[0] += (pt1 - pt0) + (1ull << 40), prof_u[1] += pt2 - pt1; // the step behind a flush of records (count << 40 | directory wait; slot wait)
						if (prof_last) prof_t[4] += pt0 - prof_last; // (from the end of one common step to the start of the next: the test, the loop)
						prof_last = pt3;
					}
#endif
#ifdef RB3_EXP_VALU /* kernel experiment: what do RB3_EXP_VALU more vector instructions per iteration cost?  (four independent chains) */
					{
						uint32_t e0 = (uint32_t)kb, e1 = e0 + 1u, e2 = e0 + 2u, e3 = e0 + 3u;
#pragma unroll
						for (int q = 0; q < RB3_EXP_VALU / 4; ++q) {
							asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(e0));
							asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(e1));
							asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(e2));
							asm volatile("v_mad_u32_u24 %0, %0, 3, %0" : "+v"(e3));
						}
						if ((e0 ^ e1 ^ e2 ^ e3) == 0x7fffff1u) steps += 1000000u; // (keeps the chains alive; practically never true)
					}
#endif
				}
			}
#endif
#ifdef RB3_PROF_STEP
			prof_last = 0;
			const uint64_t ptg = __builtin_amdgcn_s_memtime(); // (the general step: its time and count, packed, in prof_t[6])
#endif
			++it;
			pf_g = 0x80000000u;
			bool met = TEXT ? (int64_t)rc >= 0 : (int64_t)x >= 0;  // this row already carries a record
#ifdef RB3GPU_TEST_HOOKS
			if (TENT && hide_first && met && gap == 0) {
				const int64_t seen0 = TEXT ? (int64_t)rc : (int64_t)x;
				if (seen0 & RB3_TENT) {
					const int idh = (int)(seen0 >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1);
					if (idh != RB3_TENT_POISON && (__hip_atomic_load(&tab[idh].w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0ull) met = false;
				}
			}
#endif
			const int c = (int)(x & 7u);
			const int64_t kbn = TEXT ? (int64_t)(x1 >> 3) : met ? kb : RB3_ROW_NEXT(x);
			const int64_t tpn = tp > 0 ? tp - 1 : 0;
			const bool wide = TENT ? gap == 2 : gap != 0;
			// may this walker record tentatively?  (an interval of at most KMAX rows, and old enough)
			const bool tentok = TENT && gap != 0 && age >= (LIST ? RB3_TENT_MIN_AGE : RB3_TENT_MIN_AGE_AUTO) && hi - lo <= kmax && sid != -2;
			if (TENT && LIST) { // steps of walkers old enough to record whose interval is wider than the masks take: misc[39] (MISC_WIDE), see merge_core (rare: counted where it happens, not in a register)
				const unsigned long long mw = __builtin_amdgcn_ballot_w64(j == 0 && gap == 2 && sid == -1 && age >= RB3_TENT_MIN_AGE && hi - lo > kmax && hi - lo <= RB3_TENT_KMAX_TOP);
				if (mw != 0ull && lane == (int)__builtin_ctzll(mw)) atomicAdd(nsteps + 38, (unsigned long long)__builtin_popcountll(mw));
			}
			RankLoadC rl, rh;
			octc_issue_grp<DENSE, LPW>(b1, lo, c, j, rl);
			if (wide) octc_issue_grp<DENSE, LPW>(b1, hi, c, j, rh);
			uint64_t xn = 0, rcn = ~0ull;
			if (TEXT) {
				if (TEXT == 1) xn = tw[tpn > 0 ? tpn - 1 : 0]; // the word after next
				else if ((it & (uint32_t)(LPW - 1)) == 0) { // a new window: lane j fetches the word after next of phase j
					const int64_t a = tp - 2 - j;
					blk8 = tw[a > 0 ? a : 0];
				}
				if (remaining == 1 || remaining > RB3_BEYOND) rcn = (uint64_t)ld_pos(&row[trec ? tpn : kbn]);
			} else xn = (uint64_t)ld_pos(&row[kbn]);
			bool end_next;
			if (LIST) end_next = remaining == 1;
			else end_next = M && kbn >= m2 && (kbn & (M - 1)) == 0;
			octc_issue_slot<DENSE, LPW>(b1, j, rl);
			if (wide) octc_issue_slot_hi<DENSE, LPW>(b1, j, rh, rl);
			// this row: record it unless somebody already has
			++steps;
			const bool fin = met || c == 0;
			const int64_t myval = lo + kb;
			if (TENT && tentok && !met && sid < 0) { // first tentative record of this walker: open a stretch (rare)
				uint32_t s0 = 0;
				if (j == 0) s0 = gap == 1 ? atomicAdd(sidctr + 1, 1u) : tent_take_chunk(sidctr, mctr, myctr);
				s0 = grp_bcast0<LPW>(s0, j);
				if (gap == 1) sid = s0 < lim_singles ? (int)(RB3_TENT_HALF + s0) : -2; // table full: this walker stays a plain inexact one
				else sid = s0 + RB3_TENT_CHUNK <= lim_blocks ? (int)s0 : -2;
				sid0 = sid;
			}
			if (TENT && met) { // settle an unknown (rare)
				const int64_t seen = TEXT ? (int64_t)rc : (int64_t)x;
				if (TEXT && LIST && jmet != nullptr && j == 0) jmet[wcur] = tp + 1; // a junction: this row's record is somebody else's, the row before it (text position tp + 1) mine
				if (!(seen & RB3_TENT)) { // a final value: the unknown of my current stretch
					if (gap != 0 && sid >= 0 && sid != RB3_TENT_POISON && j == 0) tab[sid].del = 1 + (int)(seen - myval);
				} else {
					const int id2 = (int)(seen >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1);
					const int64_t diff = myval - (seen & RB3_TENT_MASK); // both intervals contain ka
					if (j == 0 && id2 != RB3_TENT_POISON) {
						if (gap == 0) tab[id2].del = 1 + (int)diff;
						else if (sid >= 0 && id2 != sid) tab[id2].w0 = RB3_DEP_W0(RB3_DEP_LINK, sid, 0), tab[id2].w1 = (uint64_t)(uint32_t)(int32_t)diff, tab[sid].child = id2 + 1;
					}
				}
			}
			if ((gap == 0 || (tentok && sid >= 0)) && !met && j == (int)(it & (uint32_t)(LPW - 1)))
				bkb = (TEXT && trec) ? tp : kb, bval = gap ? (RB3_TENT | ((int64_t)sid << RB3_TENT_PBITS) | myval) : myval;
			if ((it & (uint32_t)(LPW - 1)) == (uint32_t)(LPW - 1)) { // the end of a window of eight iterations: its events (noted by the common step), then its records
				evq_flush();
				if (bkb >= 0) { rec_pos<TENT && !LIST>(&row[bkb], bval, vis); bkb = -1; }
			}
			// next insertion point(s)
			uint32_t match = 0, mh;
			int64_t lo_n, hi_n;
			// A lower bound in a run slot -- the usual state in an index that holds many relatives -- goes through ONE packed decode that
			// also yields the upper bound if that lies in the same slot, and an EXACT walker is the interval [lo, lo) of the same code:
			// once a walker has passed a variant private to the new string it stays exact for the rest of its life, so some group of
			// almost every wave is exact, and giving those their own single-bound decode made 80 % of the wave iterations run both
			// sides of this branch (measured, K = 40: 71 % had an exact group, 21 % an interval over two slots, 1 % a one-row interval).
			// Left for the general side: one-row intervals (they need the match bit), bit-plane slots, the second slot of an interval.
			const uint32_t hdr0q = DENSE ? 0u : grp_bcast0<LPW>(rl.sl.x, j);
			const bool same = wide && rh.sidx == rl.sidx;
			const bool pairable = !DENSE && (hdr0q & RB3_SLOT_RLE) && (!TENT || gap != 1);
#ifdef RB3_PROF
			if (__ballot(!pairable || (wide && !same)) != 0ull) ++prof_nonpair;
#endif
			if (pairable) octc_finish_pair<LPW>(rl, same ? rh.koff : rl.koff, hdr0q, c, j, &lo_n, &hi_n);
			else {
				lo_n = octc_finish<DENSE, LPW>(rl, c, j, &match);
				hi_n = lo_n;
				if (TENT && gap == 1) hi_n = lo_n + match;
			}
			if (wide && !(pairable && same)) hi_n = octc_finish<DENSE, LPW, false>(rh, c, j, &mh); // (no match bit wanted: the run-slot side is cheaper)
			if (TEXT == 2) { // the word after next, from the lane that fetched it (after the slot has arrived: no wait of its own)
				const int src = ((lane & ~(LPW - 1)) | (int)(it & (uint32_t)(LPW - 1))) << 2;
				xn = (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)blk8) | (uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(blk8 >> 32)) << 32;
			}
			const int64_t kn = hi_n - lo_n;
			const int gap_n = kn > 1 ? 2 : (int)kn;
			if (TENT && wide && sid >= 0 && !fin && kn >= 1 && kn < hi - lo) {
				// some of the matching suffixes are not preceded by c (once per variant among the indexed relatives):
				// the rows to come belong to a new stretch.  Only note what happened -- which rows dropped out is
				// worked out by k_events afterwards, for all events of the launch at once.
				int ns = sid + 1;
				if (sid == RB3_TENT_POISON) ns = RB3_TENT_POISON;
				else if ((ns & (RB3_TENT_CHUNK - 1)) == 0) { // this walker's chunk of ids is used up (rare)
					uint32_t s0 = 0;
					if (j == 0) s0 = tent_take_chunk(sidctr, mctr, myctr);
					s0 = grp_bcast0<LPW>(s0, j);
					// table full: the records from here on stay unsettled and the host redoes the phase
					ns = s0 + RB3_TENT_CHUNK <= lim_blocks ? (int)s0 : RB3_TENT_POISON;
				}
				if (j == 0 && ns != RB3_TENT_POISON) {
					tab[ns].w0 = RB3_DEP_W0(RB3_DEP_EVENT, sid, lo), tab[ns].w1 = (uint64_t)(hi - lo) | (uint64_t)c << 16 | (TEXT ? (uint64_t)tp << RB3_EV_TP_SHIFT : 0ull);
					tab[ns].pad[0] = (uint32_t)sid0 + 1u; // (same 64-byte record: the store rides along)
					tab[sid].child = ns + 1;
				}
				sid = ns;
			}
			++age;
			// at the end of its own segment a walker goes on only if it is exact or has tentative records out
			const bool goes_on = gap_n == 0 || (TENT && sid >= 0);
			const bool at_stop = LIST && !TEXT && kbn == stop_row; // the rest of this string is recorded on another GPU
			if (at_stop && gap_n == 0 && !fin && j == 0) st_pos(arrive, lo_n);
			active = !(fin || at_stop || (end_next && !goes_on));
			remaining = end_next ? INT64_MAX : remaining - 1;
			if (TEXT) tp = tpn, x = x1, x1 = xn, rc = rcn;
			else x = xn;
			kb = kbn, lo = lo_n, hi = hi_n, gap = gap_n;
#ifdef RB3_PROF_STEP
			asm volatile("s_nop 0" :: "v"((uint32_t)lo));
			prof_t[6] += (__builtin_amdgcn_s_memtime() - ptg) + (1ull << 32);
#endif
		} while (__all(active));
	}
	{ // one atomic per wave, not per octet (they all go to the same word, and the kernel is over when the last one has landed)
		unsigned long long tot = 0;
		for (int o = 0; o < octs; ++o) tot += (uint32_t)__builtin_amdgcn_readlane((int)steps, o * LPW);
		if (lane == 0) atomicAdd(nsteps, tot);
	}
#ifdef RB3_PROF_STEP
	if (lane == 0) for (int q = 0; q < 5; ++q) atomicAdd(nsteps + 33 + q, (unsigned long long)prof_t[q]); // misc[34..38]
	if (lane == 0) atomicAdd(nsteps + 8, (unsigned long long)prof_t[5]), atomicAdd(nsteps + 9, (unsigned long long)prof_t[6]); // misc[9], misc[10]
	if (lane == 0) atomicAdd(nsteps + 10, (unsigned long long)prof_u[0]), atomicAdd(nsteps + 11, (unsigned long long)prof_u[1]); // misc[11], misc[12]
#endif
#ifdef RB3_PROF_WAVES /* kernel experiment: where does a launch's tail come from?  One entry per wave: when it began and ended (s_memtime: one clock for the chip), where it ran (HW_ID, XCC_ID), its iterations */
	if (lane == 0) {
		const unsigned long long e = atomicAdd(&g_prof_wave_n, 1ull);
		if (e < RB3_PROF_WAVES_MAX) {
			g_prof_wave[4 * e] = prof_wt0, g_prof_wave[4 * e + 1] = __builtin_amdgcn_s_memtime();
			g_prof_wave[4 * e + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32;
			g_prof_wave[4 * e + 3] = (unsigned long long)it | (unsigned long long)steps << 32;
		}
	}
#endif
#ifdef RB3_PROF
	if (lane == 0) { // wave statistics: [8] max cycles, [9] sum cycles, [10] sum iterations, [11] waves, [12] max iterations
		const unsigned long long cyc = __builtin_readcyclecounter() - tstart;
		atomicMax(nsteps + 7, cyc); atomicAdd(nsteps + 8, cyc); atomicAdd(nsteps + 9, (unsigned long long)it); atomicAdd(nsteps + 10, 1ull); atomicAdd(nsteps + 11, prof_nonpair); // ([12]: iterations with the two-decode path)
	}
#endif
}

/* settle the unknowns of all stretches.  A stretch has at most one dependent: the next stretch of
 * the same walker (EVENT) or, for the walker's last one, the first stretch of the walker it ran
 * into (LINK) -- the dependencies form simple paths.  One group of 8 lanes per stretch that a walker
 * settled follows its path forwards until the next such stretch.  The stretches of one walker are
 * consecutive ids inside aligned blocks of 8, so the group fetches a block at a time (one record per
 * lane) and steps through it with shuffles: one memory round trip per block instead of one per stretch.
 * sfin[s] = 1 + d for every stretch that got settled, walker-settled ones included (a compact copy for
 * k_pos_finalize_check; the records are 64 bytes apart). */
__global__ void __launch_bounds__(256) k_resolve(rb3_stretch_t *tab, const uint32_t *sidctr, int32_t *sfin)
{
	const int gl = threadIdx.x & (RB3_TENT_BLOCK - 1);
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF, nb = sidctr[1] < (uint32_t)RB3_TENT_HALF ? sidctr[1] : RB3_TENT_HALF;
	for (int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / RB3_TENT_BLOCK; t < na + nb; t += ((int64_t)gridDim.x * blockDim.x) / RB3_TENT_BLOCK) {
		const int64_t i = t < na ? t : RB3_TENT_HALF + (t - na);
		const int r = tab[i].del;
		if (r <= 0) continue; // only stretches a walker settled start a path
		if (gl == 0) sfin[i] = r;
		int d = r - 1, cur = (int)i, ch = tab[i].child - 1, hops = 0;
		bool go = true;
		while (go && ch >= 0 && ch < RB3_TENT_IDS && ++hops <= RB3_TENT_IDS) { // (the hop limit only guards against a corrupt table)
			const int base = ch & ~(RB3_TENT_BLOCK - 1);
			const uint4 *rp = (const uint4*)&tab[base + gl]; // lane gl: record base + gl
			const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
			const uint32_t mw[8] = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y };
			for (int o = ch & (RB3_TENT_BLOCK - 1); ; ) {
				const uint32_t w0hi = __shfl(q0.y, o, RB3_TENT_BLOCK);
				const int del = (int)__shfl(q1.x, o, RB3_TENT_BLOCK), next = (int)__shfl(q1.y, o, RB3_TENT_BLOCK) - 1;
				if ((int)(w0hi >> (RB3_TENT_PBITS - 32) & (RB3_TENT_IDS - 1)) != cur || del != 0) { go = false; break; } // another follower's link won / settled by a walker
				// every lane applies ITS record to d; the one of lane o counts
				int nd;
				if (w0hi >> 30 == RB3_DEP_EVENT) { // the dropped rows below d no longer count
					int below = 0;
#pragma unroll
					for (int q = 0; q < 8; ++q) {
						const int tt = d - 32 * q;
						below += tt >= 32 ? __popc(mw[q]) : tt > 0 ? __popc(mw[q] & ((1u << tt) - 1u)) : 0;
					}
					nd = d - below;
				} else nd = d + (int32_t)q0.z;
				d = __shfl(nd, o, RB3_TENT_BLOCK);
				if (d < 0 || d > RB3_TENT_KMAX) { go = false; break; } // cannot be: leave it unsettled, the host redoes the phase
				if (gl == o) sfin[base + o] = d + 1;
				cur = base + o, ch = next;
				if (ch != cur + 1 || ++o == RB3_TENT_BLOCK) break; // the path leaves this block
			}
		}
	}
}

/* ---- settle, second form (round 2): one hop per WALKER instead of one per stretch -------------------------------------
 * The stretches of one walker differ by the rows that dropped out of its interval since its first stretch.  k_events gives
 * every event stretch the mask of the rows that dropped AT that event, in the coordinates of the interval as it was then
 * (index among the survivors).  k_cum turns that, walker by walker, into the CUMULATIVE mask in the coordinates of the
 * walker's first interval: the i-th survivor is the i-th zero of the cumulative mask so far.  With it every stretch follows
 * from the walker's first unknown alone, d_t = d_0 - #{dropped rows with original index < d_0}, the dependency paths run over
 * first stretches only (d_0 of the next walker = d_last of this one + the link's offset) -- 3 to 24 hops where k_resolve
 * followed up to 1400 stretches one after the other --, and all other stretches are filled in parallel (k_sfin).
 * Record fields used: pad[0] of an event stretch = 1 + the walker's first stretch (written by the walker); of a first stretch,
 * after k_cum: RB3_FIRSTFLAG | 1 + the first stretch of the walker it links into; pad[1] = 1 + the walker's last stretch; mask[] of
 * a first stretch = the cumulative mask of the last one. */
#define RB3_FIRSTFLAG 0x80000000u
#define RB3_RESW_MAXHOPS 64

/* position of the n-th (0-based) set bit of z */
__device__ __forceinline__ int select32(uint32_t z, int n)
{
	int pos = 0;
	uint32_t t;
	t = __popc(z & 0xFFFFu); if (n >= (int)t) n -= (int)t, pos += 16, z >>= 16;
	t = __popc(z & 0xFFu);   if (n >= (int)t) n -= (int)t, pos += 8, z >>= 8;
	t = __popc(z & 0xFu);    if (n >= (int)t) n -= (int)t, pos += 4, z >>= 4;
	t = __popc(z & 0x3u);    if (n >= (int)t) n -= (int)t, pos += 2, z >>= 2;
	t = z & 1u;              if (n >= (int)t) pos += 1;
	return pos;
}

__device__ __forceinline__ uint32_t oct_max(uint32_t v)
{
	uint32_t t;
	t = dpp_mov<0x141>(v); v = v > t ? v : t;
	t = dpp_mov<0xB1>(v);  v = v > t ? v : t;
	t = dpp_mov<0x4E>(v);  v = v > t ? v : t;
	return v;
}

#ifdef RB3_DEBUG_CUM
__device__ unsigned long long g_cum_dbg[8];
#define RB3_CUM_DBG(i) do { if (j == 0) atomicAdd(&g_cum_dbg[i], 1ull); } while (0)
#else
#define RB3_CUM_DBG(i) do {} while (0)
#endif
/* one octet per first stretch of a walker that has stretch-id blocks (ids 0 .. na-1, multiples of 8 with pad[0] == 0) */
__global__ void __launch_bounds__(256) k_cum(rb3_stretch_t *tab, const uint32_t *sidctr)
{
	__shared__ uint32_t cumL_[32][8];        // the cumulative mask of the octet's walker
	__shared__ uint32_t blk_[32][8][8];      // the masks of the 8 records of the block being stepped through
	const int j = threadIdx.x & 7, oi = threadIdx.x >> 3;
	uint32_t *cumL = cumL_[oi];
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF;
	for (int64_t F = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3) * RB3_TENT_BLOCK; F < na; F += (((int64_t)gridDim.x * blockDim.x) >> 3) * RB3_TENT_BLOCK) {
		if (tab[F].pad[0] != 0u) continue;   // the block continues another walker's stretches
		int next = tab[F].child - 1;
		if (next < 0) continue;              // a walker without events that nobody links out of
		bool want_first = tab[F].del == 0;   // nobody settled the first stretch itself: a del on a later one may (below)
		RB3_CUM_DBG(0);
		if (want_first) RB3_CUM_DBG(1);
		cumL[j] = 0u;
		int cur = (int)F;
		bool go = true;
		int hops = 0;
		while (go && next >= 0 && next < RB3_TENT_HALF && ++hops <= RB3_TENT_IDS) {
			const int base = next & ~(RB3_TENT_BLOCK - 1);
			const uint4 *rp = (const uint4*)&tab[base + j]; // lane j: record base + j
			const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
			{
				uint32_t *bw = blk_[oi][j];
				bw[0] = q1.z, bw[1] = q1.w, bw[2] = q2.x, bw[3] = q2.y, bw[4] = q2.z, bw[5] = q2.w, bw[6] = q3.x, bw[7] = q3.y;
			}
			wave_sync();
			for (int o = next & (RB3_TENT_BLOCK - 1); ; ) {
				const uint32_t w0hi = __shfl(q0.y, o, RB3_TENT_BLOCK);
				const int nxt = (int)__shfl(q1.y, o, RB3_TENT_BLOCK) - 1;
				if (w0hi >> 30 != RB3_DEP_EVENT || (int)(w0hi >> (RB3_TENT_PBITS - 32) & (RB3_TENT_IDS - 1)) != cur) { go = false; break; } // not this walker's next event
				// the rows that dropped at this event, from survivor indices to the coordinates of the first interval
				const uint32_t cw = cumL[j];
				const uint32_t zc = 32u - __popc(cw), Z = oct_exscan(zc, j);
				uint32_t m = blk_[oi][o][j];
				const uint32_t most = oct_max((uint32_t)__popc(m));
				for (uint32_t it = 0; it < most; ++it) { // (uniform over the octet: the shuffles below need every lane)
					const bool act = m != 0u;
					const int p = act ? __ffs(m) - 1 : 0;
					m &= m - 1u;
					const uint32_t i = 32u * (uint32_t)j + (uint32_t)p; // index among the survivors
					int jj = 0;
#pragma unroll
					for (int l = 1; l < 8; ++l) jj += __shfl(Z, l, 8) <= i ? 1 : 0;
					const uint32_t Zj = __shfl(Z, jj, 8), cj = __shfl(cw, jj, 8);
					if (act) atomicOr(&cumL[jj], 1u << select32(~cj, (int)(i - Zj)));
				}
				wave_sync();
				tab[base + o].mask[j] = cumL[j]; // cumulative, first-interval coordinates
				// A follower that knew more may have settled THIS stretch (del) without ever seeing a record of the first one -- it ran so
				// closely behind the walker that those records were not visible yet, and recorded over them.  d_t, the unknown here, counts
				// the survivors below the new suffix; the unknown d_0 of the first stretch is a position whose number of survivors (zeros of
				// the cumulative mask) below it is d_t: every position from behind the (d_t - 1)-th survivor to the d_t-th survivor itself.
				// All of them give the same unknowns for this and every later stretch, but not for the earlier ones.  So the smallest one
				// becomes the walker's d_0 (del of the first stretch), and unless the range is ONE position (no dropped row just below the
				// d_t-th survivor) the first stretch's `child` -- no longer needed once this loop has passed it -- is replaced by
				// -(t + 2): "d_0 only holds from stretch t on".  k_resolve_w then starts a path here without settling the first stretch
				// itself, k_sfin settles the stretches from t on; records of earlier stretches, if any survived the follower's own
				// records, stay unsettled and the merge is redone (correctness never rests on this).
				if (want_first) {
					const int dt = (int)__shfl(q1.x, o, RB3_TENT_BLOCK) - 1; // del - 1 of this stretch
					if (dt >= 0) {
						want_first = false;
						RB3_CUM_DBG(2);
						const uint32_t cw2 = cumL[j];
						const uint32_t zc2 = 32u - __popc(cw2), Z2 = oct_exscan(zc2, j);
						const uint32_t tz = oct_sum(zc2);
						if ((uint32_t)dt < tz) { // (the d_t-th survivor exists)
							int j1 = 0, j0 = 0;
#pragma unroll
							for (int l = 1; l < 8; ++l) {
								const uint32_t Zl = __shfl(Z2, l, 8);
								j1 += Zl <= (uint32_t)dt ? 1 : 0, j0 += (dt > 0 && Zl <= (uint32_t)(dt - 1)) ? 1 : 0;
							}
							const int pmax = 32 * j1 + select32(~__shfl(cw2, j1, 8), dt - (int)__shfl(Z2, j1, 8));
							const int pmin = dt > 0 ? 32 * j0 + select32(~__shfl(cw2, j0, 8), dt - 1 - (int)__shfl(Z2, j0, 8)) + 1 : 0;
							if (pmin == pmax) RB3_CUM_DBG(3); else RB3_CUM_DBG(4);
							if (j == 0) {
								tab[F].del = 1 + pmin;
								if (pmin != pmax) tab[F].child = -(base + o) - 2;
							}
						}
					}
				}
				cur = base + o, next = nxt;
				if (next != cur + 1 || ++o == RB3_TENT_BLOCK) break; // the sequence leaves this block
			}
			wave_sync();
		}
		// summary in the first stretch: total drops, last stretch, and the first stretch of the walker this one linked into
		tab[F].mask[j] = cumL[j];
		if (j == 0) {
			uint32_t nf = 0u;
			if (next >= 0 && next < RB3_TENT_IDS) {
				const uint64_t w0 = tab[next].w0;
				if (w0 >> 62 == RB3_DEP_LINK && RB3_DEP_PREV(w0) == cur) nf = (uint32_t)next + 1u;
			}
			tab[F].pad[1] = (uint32_t)cur + 1u;
			tab[F].pad[0] = RB3_FIRSTFLAG | nf;
		}
		wave_sync();
	}
}

__device__ __forceinline__ int popc_below256(const uint32_t m[8], int d)
{
	int below = 0;
#pragma unroll
	for (int q = 0; q < 8; ++q) {
		const int tt = d - 32 * q;
		below += tt >= 32 ? __popc(m[q]) : tt > 0 ? __popc(m[q] & ((1u << tt) - 1u)) : 0;
	}
	return below;
}

/* paths over first stretches: one thread per first stretch a walker settled; one record per hop */
__global__ void __launch_bounds__(256) k_resolve_w(rb3_stretch_t *tab, const uint32_t *sidctr, int32_t *sfin, unsigned long long *bad, int maxhops)
{
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF, nb = sidctr[1] < (uint32_t)RB3_TENT_HALF ? sidctr[1] : RB3_TENT_HALF;
	const int64_t nblk = (na + RB3_TENT_BLOCK - 1) / RB3_TENT_BLOCK;
	for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk + nb; t += (int64_t)gridDim.x * blockDim.x) {
		int F = (int)(t < nblk ? t * RB3_TENT_BLOCK : RB3_TENT_HALF + (t - nblk));
		const uint4 *rp = (const uint4*)&tab[F];
		uint4 q1 = rp[1];       // (del first: 99 % of the records end here, and 16 bytes are a quarter of the traffic of the whole record)
		int d0 = (int)q1.x - 1; // del
		if (d0 < 0) continue;   // only first stretches a walker settled start a path
		uint4 q0 = rp[0], q2 = rp[2], q3 = rp[3];
		if (t < nblk && q3.z != 0u && !(q3.z & RB3_FIRSTFLAG)) continue; // (a continuation block: its first id is an event stretch)
		for (int hops = 0; hops <= maxhops; ++hops) { // longer paths (a string that repeats indexed text) are left to k_wj_*
			if (d0 < 0 || d0 > RB3_TENT_KMAX) break; // cannot be: leave it unsettled, the host redoes the phase
			// (a path that STARTS at a first stretch whose unknown k_cum derived from a later stretch and could not pin down -- child < 0 --
			// does not settle the first stretch itself: see k_cum)
			if (!(hops == 0 && (q3.z & RB3_FIRSTFLAG) && (int)q1.y < 0)) sfin[F] = d0 + 1;
			int last = F, next = (int)q1.y - 1, dl = d0;
			if (q3.z & RB3_FIRSTFLAG) { // the walker had events: k_cum left the summary
				const uint32_t mw[8] = { q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y };
				dl = d0 - popc_below256(mw, d0);
				last = (int)q3.w - 1, next = (int)(q3.z & ~RB3_FIRSTFLAG) - 1;
			}
			if (next < 0 || next >= RB3_TENT_IDS) break;
			rp = (const uint4*)&tab[next];
			q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
			const uint64_t w0 = (uint64_t)q0.y << 32 | q0.x;
			if (w0 >> 62 != RB3_DEP_LINK || RB3_DEP_PREV(w0) != last || q1.x != 0u) break; // another follower's link won / settled by a walker
			d0 = dl + (int32_t)q0.z, F = next;
			if (hops == maxhops) atomicAdd(&bad[2], 1ull); // the path goes on: tell the validation pass not to bother (the host starts k_wj_*)
		}
	}
}

/* ---- long dependency paths: pointer jumping over the walkers (second chance, after k_resolve_w gave up) ----------------
 * A string that repeats indexed text end to end never makes a walker exact, so ALL its walkers hang on one path, which
 * k_resolve_w would follow hop by hop (11.5 k hops for a 4.4 Mbp genome: 5-10 ms).  The step from the first unknown of one
 * walker to the first unknown of the next is x -> x - #{dropped rows below x} + offset, and such maps compose into a map of the
 * same form (the dropped rows of the second walker are pulled back to the coordinates of the first: a select per bit; rows
 * that only the younger walker's wider interval holds lie below or above every possible x and change the offset or nothing).
 * So the path is shortened by doubling: node = first stretch, (ptr, M, c) = "my unknown is M/c applied to the unknown of ptr". */
struct WjNode {
	uint32_t M[8];
	int32_t c, ptr, val, pad; // val: >= 0 settled, -1 not yet, -2 not a node; ptr: node this one hangs on, -1 none
};

__device__ __forceinline__ int wj_id(int X, int64_t nblk) { return X < RB3_TENT_HALF ? X / RB3_TENT_BLOCK : (int)nblk + (X - RB3_TENT_HALF); }
__device__ __forceinline__ int wj_stretch(int64_t t, int64_t nblk) { return t < nblk ? (int)t * RB3_TENT_BLOCK : RB3_TENT_HALF + (int)(t - nblk); }

__global__ void __launch_bounds__(256) k_wj_init(const rb3_stretch_t *tab, const int32_t *sfin, int64_t nblk, int64_t nb, WjNode *nodes)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk + nb) return;
	const int X = wj_stretch(t, nblk);
	WjNode n;
	for (int q = 0; q < 8; ++q) n.M[q] = 0u;
	n.c = 0, n.ptr = -1, n.val = -1, n.pad = 0;
	const uint32_t h0 = tab[X].pad[0];
	if (t < nblk && h0 != 0u && !(h0 & RB3_FIRSTFLAG)) n.val = -2; // the block continues another walker's stretches
	else {
		const int s = sfin[X], del = tab[X].del;
		if (s > 0) n.val = s - 1;
		else if (del > 0 && del <= RB3_TENT_KMAX + 1) n.val = del - 1, n.pad = (t < nblk && (h0 & RB3_FIRSTFLAG) && tab[X].child < -1) ? 1 : 0; // (pad: k_cum could not pin the first stretch down: its value only serves the paths)
		else {
			const uint64_t w0 = tab[X].w0;
			if (w0 >> 62 == RB3_DEP_LINK) {
				const int pl = RB3_DEP_PREV(w0); // the LAST stretch of the walker that ran into this one
				const uint32_t hp = tab[pl].pad[0];
				const int pf = (hp == 0u || (hp & RB3_FIRSTFLAG)) ? pl : (int)hp - 1;
				if (pf >= 0 && pf < RB3_TENT_IDS && (pf >= RB3_TENT_HALF || (pf & (RB3_TENT_BLOCK - 1)) == 0)) {
					n.ptr = wj_id(pf, nblk), n.c = (int32_t)tab[X].w1;
					if (pf != pl || (tab[pf].pad[0] & RB3_FIRSTFLAG)) // the cumulative mask of its last stretch (k_cum keeps a copy in the first)
						for (int q = 0; q < 8; ++q) n.M[q] = tab[pl].mask[q];
					if (pf == pl) for (int q = 0; q < 8; ++q) n.M[q] = 0u; // it linked from its first stretch: nothing had dropped
				}
			}
		}
	}
	nodes[t] = n;
}

__device__ __forceinline__ int wj_apply(const uint32_t M[8], int c, int x) { return x - popc_below256(M, x) + c; }

__global__ void __launch_bounds__(256) k_wj_round(int64_t n, const WjNode *in, WjNode *out)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	WjNode a = in[t];
	if (a.val == -1 && a.ptr >= 0 && a.ptr < n) {
		const WjNode p = in[a.ptr];
		if (p.val >= 0) {
			const int y = wj_apply(a.M, a.c, p.val);
			a.val = y >= 0 && y <= RB3_TENT_KMAX ? y : -3; // (-3: cannot be; stays unsettled and the merge is redone)
		} else if (p.val != -1 || p.ptr < 0) a.ptr = -1; // hangs on something that will never be settled
		else { // compose: p's map first, then mine
			uint32_t M[8];
			int c = p.c + a.c, zeros = 0;
			for (int q = 0; q < 8; ++q) M[q] = p.M[q], zeros += 32 - __popc(p.M[q]);
			for (int q = 0; q < 8; ++q) {
				uint32_t m = a.M[q];
				while (m) {
					const int i = 32 * q + __ffs(m) - 1; // a row that dropped on my side, index in the coordinates p's map produces
					m &= m - 1u;
					int sidx = i - p.c;               // its rank among the rows p's first interval keeps
					if (sidx < 0) { c -= 1; continue; } // only in the wider interval, below every x: one less below the new suffix, always
					if (sidx >= zeros) continue;        // above every x
					for (int w = 0; w < 8; ++w) {       // the sidx-th zero of p.M
						const int z = 32 - __popc(p.M[w]);
						if (sidx < z) { M[w] |= 1u << select32(~p.M[w], sidx); break; }
						sidx -= z;
					}
				}
			}
			for (int q = 0; q < 8; ++q) a.M[q] = M[q];
			a.c = c, a.ptr = p.ptr;
		}
	}
	out[t] = a;
}

__global__ void __launch_bounds__(256) k_wj_apply(int64_t nblk, int64_t nb, const WjNode *nodes, int32_t *sfin)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk + nb) return;
	if (nodes[t].val >= 0 && nodes[t].pad == 0) sfin[wj_stretch(t, nblk)] = nodes[t].val + 1;
}

/* every event stretch from its walker's first unknown: sfin[s] = 1 + d_0 - #{dropped rows below d_0}; one octet per stretch */
__global__ void __launch_bounds__(256) k_sfin(const rb3_stretch_t *tab, const uint32_t *sidctr, int32_t *sfin)
{
	const int j = threadIdx.x & 7;
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF;
	for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; s < na; s += ((int64_t)gridDim.x * blockDim.x) >> 3) {
		const uint32_t head = tab[s].pad[0];
		if (head == 0u || (head & RB3_FIRSTFLAG)) continue; // a first stretch: settled by a walker or by k_resolve_w
		if (tab[s].w0 >> 62 != RB3_DEP_EVENT) continue;
		{ // settled by a walker that ran into it: exact, whatever the first stretch says
			const int dl = tab[s].del;
			if (dl != 0) { if (j == 0 && dl >= 1 && dl <= RB3_TENT_KMAX + 1) sfin[s] = dl; continue; }
		}
		int r0 = sfin[head - 1u];
		if (r0 < 1) { // the walker's first unknown is not settled -- unless k_cum derived it from a later stretch, for the stretches from that one on
			const int ch = tab[head - 1u].child, dl0 = tab[head - 1u].del;
			if (ch < -1 && dl0 >= 1 && dl0 <= RB3_TENT_KMAX + 1 && s >= (int64_t)(-ch - 2)) r0 = dl0;
			else continue;
		}
		const int d0 = r0 - 1, tt = d0 - 32 * j;
		const uint32_t mw = tab[s].mask[j];
		const uint32_t below = oct_sum(tt >= 32 ? __popc(mw) : tt > 0 ? __popc(mw & ((1u << tt) - 1u)) : 0u);
		const int d = d0 - (int)below;
		if (j == 0 && d >= 0 && d <= RB3_TENT_KMAX) sfin[s] = d + 1;
	}
}

/* ---- more than 255 matching suffixes per interval: masks of 256 Q bits, Q = 2, 4, 8 -------------------------------------
 * An index that holds K > 255 copies of a sequence gives a walker in the middle of it an interval of K rows, and with masks of
 * 256 bits such a walker cannot record anything until enough relatives have dropped out -- which, at one private variant per
 * ~1000 symbols, takes longer than its segment: the merge degrades towards one chain per string (7-13 ms per round at 400
 * relatives instead of 1.5).  The same settle with wider masks: they live in an array of their own (mx[sid * 8Q + w], ids of
 * the lower half of the table only: events and first stretches), the 64-byte records keep everything else, and the kernels
 * below are the ones above with loops over the words (word w of a mask belongs to lane w & 7 of the octet).  The host picks Q
 * from what the walkers report (k_chain counts the steps of walkers that were old enough but too wide) and starts with 1:
 * nothing changes for an index of up to 255 relatives. */

template<int Q>
__global__ void __launch_bounds__(256) k_events_x(IdxView ix, const rb3_stretch_t *tab, uint32_t *mx, const uint32_t *sidctr)
{
	constexpr int NW = 8 * Q;
	__shared__ uint32_t dmask[32][NW];
	const int j = threadIdx.x & 7;
	uint32_t *D = dmask[threadIdx.x >> 3];
	const int64_t n = *sidctr < (uint32_t)RB3_TENT_HALF ? *sidctr : RB3_TENT_HALF;
	for (int64_t sid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; sid < n; sid += ((int64_t)gridDim.x * blockDim.x) >> 3) {
		const uint64_t w0 = tab[sid].w0;
		if (w0 >> 62 != RB3_DEP_EVENT) continue;
		const uint64_t w1 = tab[sid].w1;
		const int64_t lo = (int64_t)(w0 & (uint64_t)RB3_TENT_MASK);
		const int kk = (int)(w1 & 0xFFFF), c = (int)(w1 >> 16 & 7);
#pragma unroll
		for (int q = 0; q < Q; ++q) D[q * 8 + j] = 0u;
		__builtin_amdgcn_wave_barrier();
		// the rows [lo, lo + kk) lie in the slots of lo .. lo + kk - 1: consecutive slots, of one group or two
		const int64_t last = lo + (kk > 0 ? kk - 1 : 0) < ix.n ? lo + (kk > 0 ? kk - 1 : 0) : lo;
		const int64_t g0 = lo >> RB3_GRP_BITS, g1 = last >> RB3_GRP_BITS;
		const uint32_t koff0 = (uint32_t)lo & (RB3_GRP - 1), koff1 = (uint32_t)last & (RB3_GRP - 1);
		const uint64_t sm0 = ix.grp64[g0 * 8 + 6], sm1 = ix.grp64[g1 * 8 + 6];
		const int64_t s0 = (int64_t)((uint32_t)sm0 + __popc((uint32_t)(sm0 >> 32) & ((2u << (koff0 >> RB3_WIN_BITS)) - 1u)) - 1u);
		const int64_t s1 = (int64_t)((uint32_t)sm1 + __popc((uint32_t)(sm1 >> 32) & ((2u << (koff1 >> RB3_WIN_BITS)) - 1u)) - 1u);
		const int64_t first1 = (int64_t)(uint32_t)sm1; // first slot of the second group
		for (int64_t sx = s0; sx <= s1 && sx < s0 + 4 * RB3_TENT_QMAX + 2; ++sx) {
			const uint4 sl = ix.slot16[sx * 8 + j];
			const uint32_t hdr0 = oct_bcast0(sl.x, j);
			const int64_t gs = (g1 != g0 && sx >= first1) ? g1 : g0;
			const int64_t sstart = (gs << RB3_GRP_BITS) + (int64_t)(hdr0 & 0xFFFFu);
			drops_from_slot<NW>(sl, hdr0, (int)(lo - sstart), kk, c, j, D);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int q = 0; q < Q; ++q) mx[sid * NW + q * 8 + j] = D[q * 8 + j];
		__builtin_amdgcn_wave_barrier();
	}
}

/* k_cum for wide masks: one octet per walker; the cumulative mask, the zeros before each of its words and the masks of the block
 * being stepped through live in LDS, and every dropped row finds its word by a search over the zero counts */
template<int Q>
__global__ void __launch_bounds__(64) k_cum_x(rb3_stretch_t *tab, uint32_t *mx, const uint32_t *sidctr)
{
	constexpr int NW = 8 * Q;
	__shared__ uint32_t cumO_[8][NW], cumN_[8][NW], Zw_[8][NW], blk_[8][8][NW];
	const int j = threadIdx.x & 7, oi = threadIdx.x >> 3;
	uint32_t *cumO = cumO_[oi], *cumN = cumN_[oi], *Zw = Zw_[oi];
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF;
	for (int64_t F = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3) * RB3_TENT_BLOCK; F < na; F += (((int64_t)gridDim.x * blockDim.x) >> 3) * RB3_TENT_BLOCK) {
		if (tab[F].pad[0] != 0u) continue;   // the block continues another walker's stretches
		int next = tab[F].child - 1;
		if (next < 0) continue;              // a walker without events that nobody links out of
#pragma unroll
		for (int q = 0; q < Q; ++q) cumO[q * 8 + j] = 0u, cumN[q * 8 + j] = 0u;
		bool want_first = tab[F].del == 0;   // (see k_cum: a del on a later stretch may settle the walker)
		int cur = (int)F;
		bool go = true;
		int hops = 0;
		while (go && next >= 0 && next < RB3_TENT_HALF && ++hops <= RB3_TENT_IDS) {
			const int base = next & ~(RB3_TENT_BLOCK - 1);
			const uint4 *rp = (const uint4*)&tab[base + j]; // lane j: record base + j
			const uint4 q0 = rp[0], q1 = rp[1];
			for (int t = 0; t < NW; ++t) { // the masks of the block's 8 records: 8 NW consecutive words
				const int x = t * 8 + j;
				blk_[oi][x / NW][x % NW] = mx[(int64_t)base * NW + x];
			}
			wave_sync();
			for (int o = next & (RB3_TENT_BLOCK - 1); ; ) {
				const uint32_t w0hi = __shfl(q0.y, o, RB3_TENT_BLOCK);
				const int nxt = (int)__shfl(q1.y, o, RB3_TENT_BLOCK) - 1;
				if (w0hi >> 30 != RB3_DEP_EVENT || (int)(w0hi >> (RB3_TENT_PBITS - 32) & (RB3_TENT_IDS - 1)) != cur) { go = false; break; } // not this walker's next event
				{ // zeros (survivors) before every word of the cumulative mask so far
					uint32_t zbase = 0;
#pragma unroll
					for (int q = 0; q < Q; ++q) {
						const uint32_t z = 32u - __popc(cumO[q * 8 + j]);
						Zw[q * 8 + j] = zbase + oct_exscan(z, j);
						zbase += oct_sum(z);
					}
				}
				wave_sync();
				// the rows that dropped at this event, from survivor indices to the coordinates of the first interval
#pragma unroll 1
				for (int q = 0; q < Q; ++q) {
					uint32_t m = blk_[oi][o][q * 8 + j];
					while (m) {
						const uint32_t i = 32u * (uint32_t)(q * 8 + j) + (uint32_t)(__ffs(m) - 1); // index among the survivors
						m &= m - 1u;
						int a = 0, b = NW - 1;
						while (a < b) { // the last word with Zw <= i
							const int mid = (a + b + 1) >> 1;
							if (Zw[mid] <= i) a = mid; else b = mid - 1;
						}
						atomicOr(&cumN[a], 1u << select32(~cumO[a], (int)(i - Zw[a])));
					}
				}
				wave_sync();
#pragma unroll
				for (int q = 0; q < Q; ++q) {
					const uint32_t v = cumN[q * 8 + j];
					mx[(int64_t)(base + o) * NW + q * 8 + j] = v; // cumulative, first-interval coordinates
					cumO[q * 8 + j] = v;
				}
				wave_sync();
				if (want_first) { // this stretch settled by a follower that never saw the first one: as in k_cum
					const int dt = (int)__shfl(q1.x, o, RB3_TENT_BLOCK) - 1;
					if (dt >= 0) {
						want_first = false;
						uint32_t zbase = 0;
#pragma unroll
						for (int q = 0; q < Q; ++q) { // zeros before every word of the cumulative mask INCLUDING this event
							const uint32_t z = 32u - __popc(cumO[q * 8 + j]);
							Zw[q * 8 + j] = zbase + oct_exscan(z, j);
							zbase += oct_sum(z);
						}
						wave_sync();
						if ((uint32_t)dt < zbase && j == 0) { // (the d_t-th survivor exists; one lane searches: a rare path)
							int a = 0, b = NW - 1;
							while (a < b) { const int mid = (a + b + 1) >> 1; if (Zw[mid] <= (uint32_t)dt) a = mid; else b = mid - 1; }
							const int pmax = 32 * a + select32(~cumO[a], dt - (int)Zw[a]);
							int pmin = 0;
							if (dt > 0) {
								a = 0, b = NW - 1;
								while (a < b) { const int mid = (a + b + 1) >> 1; if (Zw[mid] <= (uint32_t)(dt - 1)) a = mid; else b = mid - 1; }
								pmin = 32 * a + select32(~cumO[a], dt - 1 - (int)Zw[a]) + 1;
							}
							tab[F].del = 1 + pmin;
							if (pmin != pmax) tab[F].child = -(base + o) - 2;
						}
						wave_sync();
					}
				}
				cur = base + o, next = nxt;
				if (next != cur + 1 || ++o == RB3_TENT_BLOCK) break; // the sequence leaves this block
			}
			wave_sync();
		}
		// summary in the first stretch: total drops, last stretch, and the first stretch of the walker this one linked into
#pragma unroll
		for (int q = 0; q < Q; ++q) mx[F * NW + q * 8 + j] = cumO[q * 8 + j];
		if (j == 0) {
			uint32_t nf = 0u;
			if (next >= 0 && next < RB3_TENT_IDS) {
				const uint64_t w0 = tab[next].w0;
				if (w0 >> 62 == RB3_DEP_LINK && RB3_DEP_PREV(w0) == cur) nf = (uint32_t)next + 1u;
			}
			tab[F].pad[1] = (uint32_t)cur + 1u;
			tab[F].pad[0] = RB3_FIRSTFLAG | nf;
		}
		wave_sync();
	}
}

template<int NW>
__device__ __forceinline__ int popc_below_n(const uint32_t *m, int d)
{
	int below = 0;
	const int full = d >> 5 < NW ? d >> 5 : NW;
	for (int w = 0; w < full; ++w) below += __popc(m[w]);
	if (full < NW && (d & 31)) below += __popc(m[full] & ((1u << (d & 31)) - 1u));
	return below;
}

template<int Q>
__global__ void __launch_bounds__(256) k_resolve_w_x(rb3_stretch_t *tab, const uint32_t *mx, const uint32_t *sidctr, int32_t *sfin, unsigned long long *bad, int maxhops)
{
	constexpr int NW = 8 * Q;
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF, nb = sidctr[1] < (uint32_t)RB3_TENT_HALF ? sidctr[1] : RB3_TENT_HALF;
	const int64_t nblk = (na + RB3_TENT_BLOCK - 1) / RB3_TENT_BLOCK;
	for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk + nb; t += (int64_t)gridDim.x * blockDim.x) {
		int F = (int)(t < nblk ? t * RB3_TENT_BLOCK : RB3_TENT_HALF + (t - nblk));
		const uint4 *rp = (const uint4*)&tab[F];
		uint4 q0 = rp[0], q1 = rp[1], q3 = rp[3];
		int d0 = (int)q1.x - 1; // del
		if (d0 < 0) continue;   // only first stretches a walker settled start a path
		if (t < nblk && q3.z != 0u && !(q3.z & RB3_FIRSTFLAG)) continue; // (a continuation block: its first id is an event stretch)
		for (int hops = 0; hops <= maxhops; ++hops) {
			if (d0 < 0 || d0 > 256 * Q - 1) break; // cannot be: leave it unsettled, the host redoes the phase
			if (!(hops == 0 && (q3.z & RB3_FIRSTFLAG) && (int)q1.y < 0)) sfin[F] = d0 + 1; // (see k_resolve_w)
			int last = F, next = (int)q1.y - 1, dl = d0;
			if (q3.z & RB3_FIRSTFLAG) { // the walker had events: k_cum_x left the summary (only walkers with blocks have events: F < HALF)
				dl = d0 - popc_below_n<NW>(mx + (int64_t)F * NW, d0);
				last = (int)q3.w - 1, next = (int)(q3.z & ~RB3_FIRSTFLAG) - 1;
			}
			if (next < 0 || next >= RB3_TENT_IDS) break;
			rp = (const uint4*)&tab[next];
			q0 = rp[0], q1 = rp[1], q3 = rp[3];
			const uint64_t w0 = (uint64_t)q0.y << 32 | q0.x;
			if (w0 >> 62 != RB3_DEP_LINK || RB3_DEP_PREV(w0) != last || q1.x != 0u) break; // another follower's link won / settled by a walker
			d0 = dl + (int32_t)q0.z, F = next;
			if (hops == maxhops) atomicAdd(&bad[2], 1ull); // the path goes on: tell the validation pass not to bother (the host starts k_wj_*)
		}
	}
}

template<int Q>
__global__ void __launch_bounds__(256) k_sfin_x(const rb3_stretch_t *tab, const uint32_t *mx, const uint32_t *sidctr, int32_t *sfin)
{
	constexpr int NW = 8 * Q;
	const int j = threadIdx.x & 7;
	const int64_t na = sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : RB3_TENT_HALF;
	for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; s < na; s += ((int64_t)gridDim.x * blockDim.x) >> 3) {
		const uint32_t head = tab[s].pad[0];
		if (head == 0u || (head & RB3_FIRSTFLAG)) continue; // a first stretch: settled by a walker or by k_resolve_w_x
		if (tab[s].w0 >> 62 != RB3_DEP_EVENT) continue;
		{ // settled by a walker that ran into it: exact, whatever the first stretch says
			const int dl = tab[s].del;
			if (dl != 0) { if (j == 0 && dl >= 1 && dl <= 256 * Q) sfin[s] = dl; continue; }
		}
		int r0 = sfin[head - 1u];
		if (r0 < 1) { // (see k_sfin: the first unknown derived from a later stretch, valid from that stretch on)
			const int ch = tab[head - 1u].child, dl0 = tab[head - 1u].del;
			if (ch < -1 && dl0 >= 1 && dl0 <= 256 * Q && s >= (int64_t)(-ch - 2)) r0 = dl0;
			else continue;
		}
		const int d0 = r0 - 1;
		uint32_t below = 0;
#pragma unroll
		for (int q = 0; q < Q; ++q) {
			const int tt = d0 - 32 * (q * 8 + j);
			const uint32_t mw = mx[s * NW + q * 8 + j];
			below += tt >= 32 ? __popc(mw) : tt > 0 ? __popc(mw & ((1u << tt) - 1u)) : 0u;
		}
		const int d = d0 - (int)oct_sum(below);
		if (j == 0 && d >= 0 && d <= 256 * Q - 1) sfin[s] = d + 1;
	}
}

/* pointer jumping over the walkers (k_wj_*) with wide masks */
template<int Q>
struct WjNodeX {
	uint32_t M[8 * Q];
	int32_t c, ptr, val, pad;
};

template<int Q>
__global__ void __launch_bounds__(256) k_wj_init_x(const rb3_stretch_t *tab, const uint32_t *mx, const int32_t *sfin, int64_t nblk, int64_t nb, WjNodeX<Q> *nodes)
{
	constexpr int NW = 8 * Q;
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk + nb) return;
	const int X = wj_stretch(t, nblk);
	WjNodeX<Q> n;
	for (int q = 0; q < NW; ++q) n.M[q] = 0u;
	n.c = 0, n.ptr = -1, n.val = -1, n.pad = 0;
	const uint32_t h0 = tab[X].pad[0];
	if (t < nblk && h0 != 0u && !(h0 & RB3_FIRSTFLAG)) n.val = -2; // the block continues another walker's stretches
	else {
		const int s = sfin[X], del = tab[X].del;
		if (s > 0) n.val = s - 1;
		else if (del > 0 && del <= 256 * Q) n.val = del - 1, n.pad = (t < nblk && (h0 & RB3_FIRSTFLAG) && tab[X].child < -1) ? 1 : 0;
		else {
			const uint64_t w0 = tab[X].w0;
			if (w0 >> 62 == RB3_DEP_LINK) {
				const int pl = RB3_DEP_PREV(w0); // the LAST stretch of the walker that ran into this one
				const uint32_t hp = tab[pl].pad[0];
				const int pf = (hp == 0u || (hp & RB3_FIRSTFLAG)) ? pl : (int)hp - 1;
				if (pf >= 0 && pf < RB3_TENT_IDS && (pf >= RB3_TENT_HALF || (pf & (RB3_TENT_BLOCK - 1)) == 0)) {
					n.ptr = wj_id(pf, nblk), n.c = (int32_t)tab[X].w1;
					if (pl < RB3_TENT_HALF && (pf != pl || (tab[pf].pad[0] & RB3_FIRSTFLAG))) // the cumulative mask of its last stretch (k_cum_x keeps a copy in the first)
						for (int q = 0; q < NW; ++q) n.M[q] = mx[(int64_t)pl * NW + q];
					if (pf == pl) for (int q = 0; q < NW; ++q) n.M[q] = 0u; // it linked from its first stretch: nothing had dropped
				}
			}
		}
	}
	nodes[t] = n;
}

template<int Q>
__global__ void __launch_bounds__(256) k_wj_round_x(int64_t n, const WjNodeX<Q> *in, WjNodeX<Q> *out)
{
	constexpr int NW = 8 * Q;
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	WjNodeX<Q> a = in[t];
	if (a.val == -1 && a.ptr >= 0 && a.ptr < n) {
		const WjNodeX<Q> &p = in[a.ptr];
		const int pval = p.val, pptr = p.ptr, pc = p.c;
		if (pval >= 0) {
			const int y = pval - popc_below_n<NW>(a.M, pval) + a.c;
			a.val = y >= 0 && y <= 256 * Q - 1 ? y : -3; // (-3: cannot be; stays unsettled and the merge is redone)
		} else if (pval != -1 || pptr < 0) a.ptr = -1; // hangs on something that will never be settled
		else { // compose: p's map first, then mine
			uint32_t M[NW];
			int c = pc + a.c, zeros = 0;
			for (int q = 0; q < NW; ++q) M[q] = p.M[q], zeros += 32 - __popc(p.M[q]);
			for (int q = 0; q < NW; ++q) {
				uint32_t m = a.M[q];
				while (m) {
					const int i = 32 * q + __ffs(m) - 1; // a row that dropped on my side, index in the coordinates p's map produces
					m &= m - 1u;
					int sidx = i - pc;                // its rank among the rows p's first interval keeps
					if (sidx < 0) { c -= 1; continue; } // only in the wider interval, below every x: one less below the new suffix, always
					if (sidx >= zeros) continue;        // above every x
					for (int w = 0; w < NW; ++w) {      // the sidx-th zero of p.M
						const uint32_t pw = p.M[w];
						const int z = 32 - __popc(pw);
						if (sidx < z) { M[w] |= 1u << select32(~pw, sidx); break; }
						sidx -= z;
					}
				}
			}
			for (int q = 0; q < NW; ++q) a.M[q] = M[q];
			a.c = c, a.ptr = pptr;
		}
	}
	out[t] = a;
}

template<int Q>
__global__ void __launch_bounds__(256) k_wj_apply_x(int64_t nblk, int64_t nb, const WjNodeX<Q> *nodes, int32_t *sfin)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nblk + nb) return;
	if (nodes[t].val >= 0 && nodes[t].pad == 0) sfin[wj_stretch(t, nblk)] = nodes[t].val + 1;
}

/* after the chains: rewrite tentative records (pos = lo + bit + kb), then every row must be recorded
 * and pos must be strictly increasing (ka is non-decreasing in kb).
 * bad[0] += #unset, bad[1] += #order violations, bad[2] += #unsettled tentative records */
__device__ __forceinline__ int64_t pos_final(int64_t v, const int32_t *sfin, unsigned long long *bad)
{
	if (v < 0) return RB3_UNSET; // never visited (still an LF word)
	if (!(v & RB3_TENT)) return v;
	const int r = sfin[(int)(v >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1)];
	if (r < 1 || r > RB3_TENT_KMAX_TOP + 1) { if (bad) atomicAdd(&bad[2], 1ull); return RB3_UNSET; }
	return (v & RB3_TENT_MASK) + (r - 1);
}

/* bad[2]: written by the settle kernels BEFORE this launch on the same stream, so an ordinary load sees it (and every compute unit finds
 * it in its L2 after the first one asked).  It used to be a volatile load: 65 k waves asking the memory side for ONE word, one after the
 * other -- most of what the pass took. */
#ifdef RB3_EXP_VOLBAD /* kernel experiment: the old load */
#define RB3_SETTLE_INCOMPLETE(bad) (*(volatile unsigned long long*)&(bad)[2] != 0)
#else
#define RB3_SETTLE_INCOMPLETE(bad) ((bad)[2] != 0)
#endif
__global__ void __launch_bounds__(256) k_pos_finalize(int64_t *pos, int64_t n2, const int32_t *sfin, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n2) return;
	const int64_t v = pos[i];
	if (v >= 0 && (v & RB3_TENT)) pos[i] = pos_final(v, sfin, bad);
}

/* both in one pass over pos[] (each thread finalises its own row and re-derives its left neighbour) */
__global__ void __launch_bounds__(256) k_pos_finalize_check(int64_t *pos, int64_t n2, int64_t ntot, const int32_t *sfin, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n2) return;
	if (RB3_SETTLE_INCOMPLETE(bad)) return; // the settle pass already knows it is incomplete: nothing to validate yet
	const int64_t raw = pos[i];
	const int64_t p = pos_final(raw, sfin, bad);
	const int64_t q = i > 0 ? pos_final(pos[i - 1], sfin, nullptr) : RB3_UNSET; // the neighbour's own thread reports its problems
	if (p != raw && p >= 0) pos[i] = p; // (an unsettled record stays as it is: a longer settle pass may still resolve it)
	if (p < 0) atomicAdd(&bad[0], 1ull);
	else if (p >= ntot || (i > 0 && q >= 0 && q >= p)) atomicAdd(&bad[1], 1ull);
}

/* the same plus k_win_rows in the same pass (single-sync merge with the window-parallel rebuild): thread i also
 * writes jw[w] = i for every output window w that begins between row i-1 and row i (thread n2: the windows
 * after the last row).  Rows that failed validation write nothing -- the rebuild is skipped then anyway. */
template<bool TENT>
__global__ void __launch_bounds__(256) k_pos_finalize_check_rows(int64_t *pos, int64_t n2, int64_t ntot, const int32_t *sfin, unsigned long long *bad,
		int64_t *jw, int64_t nwin, const int64_t *rec = nullptr, const uint32_t *sa = nullptr)
{
	// rec, sa: the walkers left their records in TEXT order (k_chain, trec): row i's record is rec[sa[i]], and pos[] is written here
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n2) return;
	// the settle pass already knows it is incomplete: nothing to validate yet (one answer per wave: the lanes exchange values below)
	if (TENT && __builtin_amdgcn_readfirstlane((int)(RB3_SETTLE_INCOMPLETE(bad))) != 0) return;
	int64_t p = INT64_MAX, q = RB3_UNSET;
	if (i < n2) {
		const int64_t raw = sa ? rec[sa[i] < (uint32_t)n2 ? sa[i] : 0u] : pos[i];
		p = TENT ? pos_final(raw, sfin, bad) : (raw < 0 ? RB3_UNSET : raw);
		if (sa) pos[i] = p; // (an unsettled record stays in rec[]: a longer settle pass may still resolve it)
		else if (TENT && p != raw && p >= 0) pos[i] = p; // (an unsettled record stays as it is: a longer settle pass may still resolve it)
	}
	// the row before: from the lane below (its final value is in a register there); only lane 0 of a wave looks it up again
	// (the neighbour's own thread reports its problems)
	{
		const uint32_t qlo = wave_up1((uint32_t)(uint64_t)p), qhi = wave_up1((uint32_t)((uint64_t)p >> 32));
		if ((threadIdx.x & 63) != 0) q = (int64_t)((uint64_t)qhi << 32 | qlo);
		else if (i > 0) {
			const int64_t rq = sa ? rec[sa[i - 1] < (uint32_t)n2 ? sa[i - 1] : 0u] : pos[i - 1];
			q = TENT ? pos_final(rq, sfin, nullptr) : (rq < 0 ? RB3_UNSET : rq);
		}
	}
	bool ok = true;
	if (i < n2) {
		if (p < 0) { atomicAdd(&bad[0], 1ull); ok = false; }
		else if (p >= ntot || (i > 0 && q >= 0 && q >= p)) { atomicAdd(&bad[1], 1ull); ok = false; }
	}
	if (i > 0 && q < 0) ok = false;
	if (!ok) return;
	const int64_t a = i == 0 ? -1 : q >> RB3_WIN_BITS;
	int64_t b = i == n2 ? nwin : p >> RB3_WIN_BITS;
	if (b > nwin) b = nwin;
	for (int64_t w = a + 1; w <= b; ++w) jw[w] = i;
}

/* the same for records in row order (rec == nullptr above), R rows per thread: 16-byte loads and stores.  The pass is a stream of 16 B per
 * row plus one gather from sfin[] per tentative row (8.8 M rows: 163 MB and 8.5 M gathers that hit the L2, 66 us = 2.5 TB/s).  Two rows per
 * thread since round 4 (one: half the rate); four were measured in round 5 and are 7 us per launch SLOWER (profiles/r5_ab_finalize_rows.txt). */
#ifndef RB3_FIN_ROWS
#define RB3_FIN_ROWS 2
#endif
template<bool TENT, int R>
__global__ void __launch_bounds__(256) k_pos_finalize_check_rowsN(int64_t *pos, int64_t n2, int64_t ntot, const int32_t *sfin, unsigned long long *bad, int64_t *jw, int64_t nwin)
{
	static_assert(R == 2 || R == 4, "two or four rows per thread");
	const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * R;
	if (i0 > n2) return;
	if (TENT && __builtin_amdgcn_readfirstlane((int)(RB3_SETTLE_INCOMPLETE(bad))) != 0) return;
	int64_t raw[R], p[R], q = RB3_UNSET;
	const bool full = i0 + R <= n2;
	if (full) {
#pragma unroll
		for (int r = 0; r < R; r += 2) {
			const longlong2 v = *(const longlong2*)(pos + i0 + r);
			raw[r] = v.x, raw[r + 1] = v.y;
		}
	} else {
#pragma unroll
		for (int r = 0; r < R; ++r) raw[r] = i0 + r < n2 ? pos[i0 + r] : 0;
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		p[r] = INT64_MAX;
		if (full || i0 + r < n2) p[r] = TENT ? pos_final(raw[r], sfin, bad) : (raw[r] < 0 ? RB3_UNSET : raw[r]);
	}
	if (TENT) { // (an unsettled record stays as it is: a longer settle pass may still resolve it)
		if (full) {
#pragma unroll
			for (int r = 0; r < R; r += 2)
				if ((p[r] != raw[r] && p[r] >= 0) || (p[r + 1] != raw[r + 1] && p[r + 1] >= 0)) {
					longlong2 w;
					w.x = p[r] >= 0 ? p[r] : raw[r], w.y = p[r + 1] >= 0 ? p[r + 1] : raw[r + 1];
					*(longlong2*)(pos + i0 + r) = w;
				}
		} else {
#pragma unroll
			for (int r = 0; r < R; ++r)
				if (i0 + r < n2 && p[r] != raw[r] && p[r] >= 0) pos[i0 + r] = p[r];
		}
	}
	{ // the row before i0: the last row of the lane below (all of its rows exist, since this thread's first one does or is row n2); lane 0 of a wave looks it up again
		const int64_t last = p[R - 1];
		const uint32_t qlo = wave_up1((uint32_t)(uint64_t)last), qhi = wave_up1((uint32_t)((uint64_t)last >> 32));
		if ((threadIdx.x & 63) != 0) q = (int64_t)((uint64_t)qhi << 32 | qlo);
		else if (i0 > 0) {
			const int64_t rq = pos[i0 - 1];
			q = TENT ? pos_final(rq, sfin, nullptr) : (rq < 0 ? RB3_UNSET : rq); // (the neighbour's own thread reports its problems)
		}
	}
#pragma unroll
	for (int r = 0; r < R; ++r) {
		const int64_t i = i0 + r; // row i, or, with i == n2, the windows behind the last row
		if (i > n2) break;
		const int64_t prev = r == 0 ? q : p[r - 1];
		bool ok = true;
		if (i < n2) {
			if (p[r] < 0) { atomicAdd(&bad[0], 1ull); ok = false; }
			else if (p[r] >= ntot || (i > 0 && prev >= 0 && prev >= p[r])) { atomicAdd(&bad[1], 1ull); ok = false; }
		}
		if (i > 0 && prev < 0) ok = false;
		if (ok) {
			const int64_t a = i == 0 ? -1 : prev >> RB3_WIN_BITS;
			int64_t b = i == n2 ? nwin : p[r] >> RB3_WIN_BITS;
			if (b > nwin) b = nwin;
			for (int64_t w = a + 1; w <= b; ++w) jw[w] = i;
		}
	}
}

__global__ void __launch_bounds__(256) k_pos_check(const int64_t *pos, int64_t n2, int64_t ntot, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n2) return;
	const int64_t p = pos[i];
	if (p < 0) atomicAdd(&bad[0], 1ull); // never visited
	else if (p >= ntot || (i > 0 && pos[i - 1] >= 0 && pos[i - 1] >= p)) atomicAdd(&bad[1], 1ull);
}

#ifdef RB3GPU_TEST_HOOKS
/* test hook: a wrong-but-monotone pos[] -- every row in [n2/3, n2/2) whose successor leaves room moves one position up */
__global__ void __launch_bounds__(256) k_test_corrupt(int64_t *pos, int64_t n2)
{
	const int64_t i = n2 / 3 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i + 1 >= n2 / 2) return;
	const int64_t p = pos[i], q = pos[i + 1];
	if (p >= 0 && q >= 0 && !(p & RB3_TENT) && !(q & RB3_TENT) && q - p >= 3) pos[i] = p + 1;
}
#endif

/* Sampled check of pos[] against the index itself, independent of how the walkers got there: for a
 * row kb of the batch with symbol c != $, the row of the suffix one symbol longer is kb' = C2[c] + rank_B2(c, kb) and its
 * insertion point must be ka[kb'] = C1[c] + rank_B1(c, ka[kb]) (fm-index.c:171-173), where ka[r] = pos[r] - r; the sentinel
 * rows have ka = m1 (fm-index.c:164).  If this holds for every row, pos[] is the reference's rb[] >> 6 by induction along
 * every string; here it is verified for every `stride`-th row (one octet per sample: the count inside the 4096-byte tile of
 * B2, then one rank on B1), which finds any systematic error of the speculative walkers at sizes no CPU oracle reaches.
 * nchecked[1] counts mismatches (a merge that fails here is redone without speculative records, then reported). */
__global__ void __launch_bounds__(256) k_lf_check(IdxView b1, const int64_t *pos, const uint8_t *b2, int64_t n2, const uint64_t *tpre, const uint64_t *tot2,
		int64_t stride, unsigned long long *bad, unsigned long long *nchecked)
{
	if ((bad[0] | bad[1] | bad[2]) != 0) return; // pos[] already failed validation
	const int lane = threadIdx.x & 63, j = lane & 7;
	const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	// spread the samples over the residues of the stride so that repeated merges look at different rows
	const int64_t kb = s * stride + (int64_t)((uint64_t)(s * 0x9E3779B97F4A7C15ull) >> 40) % stride;
	if (kb >= n2) return;
	const int64_t m2 = (int64_t)tot2[0];
	const int64_t ka = pos[kb] - kb;
	const int c = (int)b2[kb];
	bool ok = ka >= 0 && ka <= b1.n && c <= 5;
	if (kb < m2 && ka != b1.m) ok = false; // a sentinel row
	if (ok && c != 0) {
		const int64_t tile = kb >> 12, base = tile << 12;
		const int r = (int)(kb - base);
		uint32_t cnt = 0;
		if (((uintptr_t)b2 & 7) == 0) { // eight bytes at a time: bytes are 0..7, so (y + 0x7f) has bit 7 set exactly in the bytes that differ from c
			const uint64_t *w8 = (const uint64_t*)(b2 + base);
			const uint64_t pat = 0x0101010101010101ull * (uint64_t)c;
			for (int w = j; w * 8 < r; w += 8) {
				const uint64_t y = w8[w] ^ pat;
				uint64_t ne = ((y & 0x0707070707070707ull) + 0x7F7F7F7F7F7F7F7Full) & 0x8080808080808080ull;
				const int nbytes = r - w * 8 < 8 ? r - w * 8 : 8; // bytes of this word that lie before the row
				if (nbytes < 8) ne |= ~0ull << (8 * nbytes);      // the others count as different
				cnt += 8u - (uint32_t)__popcll(ne & 0x8080808080808080ull);
			}
		} else
			for (int i = j; i < r; i += 8) cnt += b2[base + i] == (uint8_t)c ? 1u : 0u;
		cnt = oct_sum(cnt);
		int64_t c2 = 0;
		for (int a = 0; a < c; ++a) c2 += (int64_t)tot2[a];
		const int64_t kbn = c2 + (int64_t)tpre[tile * 8 + c] + cnt;
		RankLoad rl;
		oct_rank_issue(b1, ka, j, rl);
		const int64_t want = oct_rank_finish(rl, c, j, b1.abs);
		if (kbn < 0 || kbn >= n2 || pos[kbn] - kbn != want) ok = false;
	}
	if (j == 0) {
		if (!ok) atomicAdd(nchecked + 1, 1ull); // (its own counter: not to be mistaken for unsettled tentative records)
		else if ((s & 63) == 0) atomicAdd(nchecked, 64ull); // (approximate count, kept off the hot address)
	}
}

/* The DETERMINISTIC part of the validation (round 5): the LF relation at every JUNCTION of the speculative walk, i.e. at every place
 * where the value of a row was not computed from the row before it by the walker itself but put together from what two parties knew:
 *   (a) a walker met somebody's record and settled or linked an unknown there (k_chain: `met`; the text position is in jmet[walker]):
 *       the row before the met one is the walker's own, the met one is somebody else's;
 *   (b) an EVENT: matching suffixes dropped out and the rows to come belong to a new stretch whose unknown follows from the old one
 *       through the drop-out mask (k_events .. k_sfin); the text position is in the event record (w1 >> RB3_EV_TP_SHIFT).
 * Inside a stretch every record is lo + kb with lo computed by rank from the lo before it, and the final value adds the stretch's ONE
 * settled unknown to all of them: if the relation holds across the stretch's two ends it holds inside (the matching suffixes all
 * survive there, so LF maps lo + d to lo' + d for every d).  A wrong unknown of any single stretch, a wrong link, a wrong settle by a
 * follower each break the relation at a junction that is looked at here -- every time, not with the probability of a sample
 * (k_lf_check keeps sampling every 4096th row for errors of the rank arithmetic itself).  One octet per walker and per event stretch:
 * text positions t + 1 -> t, rows kb = tw[t + 1] >> 3 and kbn = tw[t] >> 3, c = tw[t + 1] & 7:  pos[kbn] - kbn == C1[c] + rank_B1(c, pos[kb] - kb)
 * (fm-index.c:171-173).  nchecked[1] counts failures (with k_lf_check's), *njunc the junctions looked at. */
#ifndef RB3_JUNC_UNROLL
#define RB3_JUNC_UNROLL 1 /* (measured, round 6: 2 / 4 / 8 junctions at a time finish the check sooner and cost the rebuild beside it more than that -- rebuild 44 -> 47 / 47.5 / 52 ms per
                             152-genome build, the whole build 167-169 -> 170-175 ms; a wider launch the same.  The check lives on latency nobody else wants.) */
#endif
__global__ void __launch_bounds__(256) k_junction_check(IdxView b1, const int64_t *pos, const uint64_t *tw, int64_t n2, const rb3_stretch_t *tab, const uint32_t *sidctr,
		const int64_t *jmet, int64_t nwalk_arg, const unsigned long long *nwalk_dev, unsigned long long *bad, unsigned long long *nchecked, unsigned long long *njunc,
		int ev_stride, int ev_phase)
{
	// ev_stride, ev_phase: the events looked at are those of the stretch ids = phase (mod stride); 1: all of them (the walkers' junctions: always all)
	if ((bad[0] | bad[1] | bad[2]) != 0) return; // pos[] already failed validation
	const int lane = threadIdx.x & 63, j = lane & 7;
	const int64_t nwalk = nwalk_dev ? (int64_t)*nwalk_dev : nwalk_arg;
	const int64_t nev = (int64_t)(sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : (uint32_t)RB3_TENT_HALF); // events only happen to ids from blocks
	const int64_t nq = nwalk + (nev - ev_phase + ev_stride - 1) / ev_stride;
	unsigned long long nfail = 0, nseen = 0;
	// U junctions per octet and iteration, their loads asked for stage by stage (event record -> text words -> the two rows -> directory -> slot): a junction is a
	// chain of five round trips, and with one at a time 16 k octets needed ~350 us for the 1.2 M events of a late merge of the 152-genome build -- as long as the
	// rebuild they run beside (round 6: all events are looked at, not every 16th)
	constexpr int U = RB3_JUNC_UNROLL;
	const int64_t noct = ((int64_t)gridDim.x * blockDim.x) >> 3;
	for (int64_t q0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; q0 < nq; q0 += noct * U) {
		int64_t t[U], kb[U], kbn[U], ka[U], kan[U];
		uint64_t w0[U], w1[U], xa[U], xb[U];
		RankLoad rl[U];
		bool live[U], ok[U];
#pragma unroll
		for (int u = 0; u < U; ++u) { // (junction q0 + u noct: the octets of a wave stay on neighbouring entries)
			const int64_t q = q0 + (int64_t)u * noct;
			live[u] = q < nq, t[u] = -1, w0[u] = 0, w1[u] = 0;
			if (live[u] && q < nwalk) t[u] = jmet[q] - 1;
			else if (live[u]) {
				const int64_t sid = (q - nwalk) * ev_stride + ev_phase;
				live[u] = sid < nev;
				if (live[u]) { const ulonglong2 v = *(const ulonglong2*)&tab[sid].w0; w0[u] = v.x, w1[u] = v.y; }
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t q = q0 + (int64_t)u * noct;
			if (live[u] && q >= nwalk) {
				live[u] = w0[u] >> 62 == RB3_DEP_EVENT;
				t[u] = (int64_t)(w1[u] >> RB3_EV_TP_SHIFT) - 1; // noted at text position tp: the new stretch begins at tp - 1
			}
			live[u] = live[u] && t[u] >= 0 && t[u] + 1 < n2;
			xa[u] = xb[u] = 0;
			if (live[u]) xa[u] = tw[t[u] + 1], xb[u] = tw[t[u]];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			live[u] = live[u] && (xa[u] & 7u) != 0u; // (c == 0: position t + 1 starts a string: nothing leads from it to t)
			kb[u] = (int64_t)(xa[u] >> 3), kbn[u] = (int64_t)(xb[u] >> 3);
			ok[u] = kb[u] < n2 && kbn[u] < n2;
			ka[u] = kan[u] = 0;
			if (live[u] && ok[u]) ka[u] = pos[kb[u]] - kb[u], kan[u] = pos[kbn[u]] - kbn[u];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			ok[u] = ok[u] && ka[u] >= 0 && ka[u] <= b1.n && (int)(xa[u] & 7u) <= 5;
			oct_rank_issue_grp(b1, live[u] && ok[u] ? ka[u] : 0, j, rl[u]); // (a junction that is not looked at: a valid address, the result unused)
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			oct_rank_issue_slot(b1, j, rl[u]);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int c = (int)(xa[u] & 7u);
			const bool same = oct_rank_finish(rl[u], c >= 1 && c <= 5 ? c : 1, j, b1.abs) == kan[u];
			if (live[u]) { ++nseen; if (!(ok[u] && same)) ++nfail; }
		}
	}
	if (j == 0 && nfail) atomicAdd(nchecked + 1, nfail);
	{ // ONE atomic per block for the count (every lane of an octet counted the same junctions; an atomic per wave -- 8 k of them on one word, ~12 ns
	  // of its L2 channel each -- made this kernel last 70 us, all of them in front of the rebuild)
		__shared__ unsigned int bsum;
		if (threadIdx.x == 0) bsum = 0u;
		__syncthreads();
		unsigned int tot = 0;
		for (int o = 0; o < 8; ++o) tot += (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)nseen, o * 8);
		if (lane == 0 && tot) atomicAdd(&bsum, tot);
		__syncthreads();
		if (threadIdx.x == 0 && bsum) atomicAdd(njunc, (unsigned long long)bsum);
	}
}

#ifdef RB3GPU_TEST_HOOKS
/* test hook: ONE settled stretch gets a wrong unknown -- the `which`-th event stretch in use (counted from the table's start) that is settled, moved by +1 or,
 * if that leaves the masks' range, by -1: every row of that stretch and of no other moves by one position.  what[0] = the stretch (or -1). */
__global__ void k_test_corrupt_sfin(const rb3_stretch_t *tab, const uint32_t *sidctr, int32_t *sfin, int64_t which, long long *what)
{
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	const int64_t n = (int64_t)(sidctr[0] < (uint32_t)RB3_TENT_HALF ? sidctr[0] : (uint32_t)RB3_TENT_HALF);
	int64_t seen = 0;
	what[0] = -1;
	for (int64_t s = 0; s < n; ++s) {
		if (tab[s].w0 >> 62 != RB3_DEP_EVENT || sfin[s] < 1) continue;
		if (seen++ != which) continue;
		const int kk = (int)(tab[s].w1 & 0xFFFF);
		sfin[s] += sfin[s] - 1 < kk ? 1 : -1;
		what[0] = s;
		return;
	}
}
#endif

/* ----------------------------------------------------------------------------------------- */
/* interleave + rebuild                                                                        */
/* ----------------------------------------------------------------------------------------- */

/* symbol at offset i of an existing index, one lane */
__device__ __forceinline__ uint32_t idx_sym(const IdxView &ix, int64_t i)
{
	const int64_t g = i >> RB3_GRP_BITS;
	const uint32_t koff = (uint32_t)i & (RB3_GRP - 1), lw = koff >> RB3_WIN_BITS;
	const uint64_t sm = ix.grp64[g * 8 + 6];
	const uint32_t s = (uint32_t)sm + __popc((uint32_t)(sm >> 32) & ((2u << lw) - 1u)) - 1u;
	const uint32_t *sp = (const uint32_t*)(ix.slot16 + (int64_t)s * 8);
	const uint32_t hdr0 = sp[0];
	const uint32_t off = koff - (hdr0 & 0xFFFFu);
	if (!(hdr0 & RB3_SLOT_RLE)) {
		const uint32_t jj = off >> 5, bit = off & 31;
		const uint4 sl = ix.slot16[(int64_t)s * 8 + jj];
		return ((sl.y >> bit) & 1u) | ((sl.z >> bit) & 1u) << 1 | ((sl.w >> bit) & 1u) << 2;
	} else {
		for (int q = 0; q < RB3_RLE_CODES; ++q) { // the first run that ends behind off (unused codes repeat the last end: never)
			const uint32_t word = sp[(q / 6) * 4 + 1 + (q % 6) / 2];
			const uint32_t code = (q & 1) ? word >> 16 : word & 0xFFFFu; // q%6 and q have the same parity
			if (off < RB3_RUN_END(code)) return code & 7u;
		}
		return 7;
	}
}

/* `skip`, in all rebuild kernels: the validation counters of the rank phase.  The single-sync
 * merge launches the rebuild before the host has seen them; if the rank phase failed (it will be redone) pos[]
 * is not a valid interleave and the rebuild must neither take long nor write out of bounds: it does nothing. */
#define RB3_REB_SKIP(skip) ((skip) != nullptr && ((skip)[0] | (skip)[1] | (skip)[2]) != 0)

/* jg[g] = #{rows kb : pos[kb] < g * 8192}, g = 0..ngrp  (pos is strictly increasing) */
__global__ void __launch_bounds__(256) k_group_rows(const int64_t *pos, int64_t n2, int64_t *jg, int64_t ngrp, const unsigned long long *skip)
{
	if (RB3_REB_SKIP(skip)) return;
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n2) return;
	int64_t a = i == 0 ? -1 : pos[i - 1] >> RB3_GRP_BITS; // group of the previous row
	int64_t b = i == n2 ? ngrp : pos[i] >> RB3_GRP_BITS;   // group of this row
	if (a < -1) a = -1;       // (only an invalid pos[] has negative entries; the result is discarded then,
	if (b > ngrp) b = ngrp;   //  but the loop must stay bounded)
	for (int64_t g = a + 1; g <= b; ++g) jg[g] = i;
}

/* The 256 symbols of window [p0, p0+256) of the merged BWT.  Lane t gets positions
 * p0 + 64u + t, u = 0..3, in sym[u]; 7 marks positions past the end.  `j` is the number of
 * B2 rows placed before p0 and is advanced past this window.  One wave per workgroup. */
#define RB3_REB_WAVES 4   /* waves per block of the window-parallel rebuild kernels */
#define RB3_REB_WPW   1   /* consecutive windows each of those waves handles (blocks are dispatched at ~2 G/s:
                             one 64-thread block per window would be bound by that) */

template<bool FROM_PLAIN>
__device__ __forceinline__ void gen_window(const IdxView &old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot,
		int64_t p0, int64_t &j, uint8_t *symbuf, uint32_t sym[4], int lane)
{
	if (FROM_PLAIN) {
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int64_t p = p0 + 64 * u + lane;
			sym[u] = p < ntot ? b2[p] : 7u;
		}
		return;
	}
	// the old symbols of the window are consecutive positions of the old index, at most 256 of them: they
	// lie in at most two of its windows, i.e. two slots, which the wave copies to LDS once (run slots:
	// plus the start offset of every run) instead of walking directory and slot once per symbol
	__shared__ uint32_t oslot_[RB3_REB_WAVES][2][32];
	__shared__ uint16_t ostart_[RB3_REB_WAVES][2][64];
	__shared__ uint8_t orsym_[RB3_REB_WAVES][2][64];
	uint32_t (*oslot)[32] = oslot_[threadIdx.x >> 6];
	uint16_t (*ostart)[64] = ostart_[threadIdx.x >> 6];
	uint8_t (*orsym)[64] = orsym_[threadIdx.x >> 6];
	const int64_t a1 = p0 - j;
	const int64_t w0 = (a1 < 0 ? 0 : a1) >> RB3_WIN_BITS;
	// round trip 1: the rows that land in this window (pos[]) and the directory entries of the two old windows
	uint64_t sm[2] = {0, 0};
	int64_t og[2] = {0, 0};
	uint32_t olw[2] = {0, 0};
	bool ohave[2];
#pragma unroll
	for (int X = 0; X < 2; ++X) {
		const int64_t wv = w0 + X;
		ohave[X] = (wv << RB3_WIN_BITS) < old.n;
		if (ohave[X]) {
			og[X] = wv >> (RB3_GRP_BITS - RB3_WIN_BITS), olw[X] = (uint32_t)wv & (RB3_GRP_WINS - 1);
			sm[X] = old.grp64[og[X] * 8 + 6];
		}
	}
	const int64_t r0 = j + lane < n2 ? pos[j + lane] : INT64_MAX;
	((uint32_t*)symbuf)[lane] = 0xFFFFFFFFu;
	wave_sync();
	// round trip 2: the two slots and the symbols of those rows
	const uint32_t *sp[2] = {nullptr, nullptr};
	uint32_t sw[2] = {0, 0}, ohdr[2] = {0, 0};
#pragma unroll
	for (int X = 0; X < 2; ++X)
		if (ohave[X]) {
			const uint32_t sidx = (uint32_t)sm[X] + __popc((uint32_t)(sm[X] >> 32) & ((2u << olw[X]) - 1u)) - 1u;
			sp[X] = (const uint32_t*)(old.slot16 + (int64_t)sidx * 8);
			sw[X] = sp[X][lane & 31];
			ohdr[X] = sp[X][0];
		}
	int nb2 = 0;
	for (int u = 0; u < 4; ++u) {
		const int64_t jj = j + 64 * u + lane;
		const int64_t r = u == 0 ? r0 : (jj < n2 ? pos[jj] : INT64_MAX);
		const bool in = r < p0 + RB3_WIN;
		if (in && r >= p0) symbuf[r - p0] = b2[jj]; // r >= p0 always holds for a valid pos[]; never write outside the window
		const uint64_t m = __ballot(in);
		nb2 += __popcll(m);
		if (m != ~0ull) break;
	}
	int64_t obase[2] = {0, 0};
#pragma unroll
	for (int X = 0; X < 2; ++X)
		if (ohave[X]) {
			if (lane < 32) oslot[X][lane] = sw[X];
			obase[X] = (og[X] << RB3_GRP_BITS) + (ohdr[X] & 0xFFFFu);
			if (ohdr[X] & RB3_SLOT_RLE) { // start offset of every run: prefix sum over the 48 codes
				uint32_t end = 0, sy = 7;
				const uint32_t word = __shfl(sw[X], lane < RB3_RLE_CODES ? (lane / 6) * 4 + 1 + (lane % 6) / 2 : 0);
				if (lane < RB3_RLE_CODES) {
					const uint32_t code = (lane & 1) ? word >> 16 : word & 0xFFFFu; // lane % 6 and lane have the same parity
					sy = code & 7u, end = RB3_RUN_END(code);
				}
				uint32_t st = wave_up1(end); // a run starts where the one before it ends (cumulative codes)
				if (lane == 0) st = 0u;
				ostart[X][lane] = (uint16_t)((lane < RB3_RLE_CODES && sy != 7u) ? st : 0xFFFFu); // unused codes start at the end
				orsym[X][lane] = (uint8_t)sy;
			}
		}
	wave_sync();
	int before = 0;
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		uint32_t s = symbuf[64 * u + lane];
		const bool isb2 = s != 0xFFu;
		const uint64_t m = __ballot(isb2);
		const int mine = before + __popcll(m & ((1ull << lane) - 1ull));
		before += __popcll(m);
		if (!isb2) {
			const int64_t p = p0 + 64 * u + lane;
			const int64_t i1 = a1 + 64 * u + lane - mine;
			s = 7u;
			if (p < ntot && i1 >= 0 && i1 < old.n) { // the range check only matters if pos[] is invalid
				const int X = (int)((i1 >> RB3_WIN_BITS) - w0) & 1;
				const uint32_t off = (uint32_t)(i1 - obase[X]);
				if (!(ohdr[X] & RB3_SLOT_RLE)) {
					const uint32_t jj = (off >> 5) & 7u, bit = off & 31u;
					s = ((oslot[X][jj * 4 + 1] >> bit) & 1u) | ((oslot[X][jj * 4 + 2] >> bit) & 1u) << 1 | ((oslot[X][jj * 4 + 3] >> bit) & 1u) << 2;
				} else { // the last run that starts at or before off
					int q = 0;
#pragma unroll
					for (int d = 32; d >= 1; d >>= 1)
						if (q + d < RB3_RLE_CODES && ostart[X][q + d] <= off) q += d;
					s = orsym[X][q];
				}
			}
		}
		sym[u] = s;
	}
	j += nb2;
	wave_sync();
}

/* run heads of a window: H[u] bit t set <=> position 64u+t starts a run */
__device__ __forceinline__ void window_heads(const uint32_t sym[4], int lane, uint64_t H[4])
{
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		uint32_t prev = wave_up1(sym[u]);
		if (lane == 0) prev = 8u;
		if (u > 0) { const uint32_t pl = wave_read(sym[u - 1], 63); if (lane == 0) prev = pl; }
		H[u] = __ballot(sym[u] != 7u && sym[u] != prev);
	}
}

/* pass 1: per group of the NEW index, symbol counts and the slot partition.
 * gstat[g*8 + 0..5] = symbol counts, [6] = #slots, [7] = slot-start mask. */
template<bool FROM_PLAIN>
__global__ void __launch_bounds__(64) k_pass1(IdxView old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot,
		const int64_t *jg, uint32_t *gstat, int64_t ngrp, const unsigned long long *skip)
{
	if (RB3_REB_SKIP(skip)) return;
	__shared__ __attribute__((aligned(16))) uint8_t symbuf[RB3_WIN];
	const int lane = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t W = (ntot >> RB3_WIN_BITS) + 1;
	const int nvw = (int)(W - g * RB3_GRP_WINS < RB3_GRP_WINS ? W - g * RB3_GRP_WINS : RB3_GRP_WINS);
	int64_t j = FROM_PLAIN ? 0 : jg[g];
	uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
	int my_nruns = 0;
	uint32_t my_first = 7, my_last = 7;
	for (int lw = 0; lw < nvw; ++lw) {
		const int64_t p0 = (g * RB3_GRP_WINS + lw) << RB3_WIN_BITS;
		uint32_t sym[4];
		gen_window<FROM_PLAIN>(old, pos, b2, n2, ntot, p0, j, symbuf, sym, lane);
		uint64_t H[4];
		window_heads(sym, lane, H);
		const int nruns = __popcll(H[0]) + __popcll(H[1]) + __popcll(H[2]) + __popcll(H[3]);
#pragma unroll
		for (int u = 0; u < 4; ++u)
#pragma unroll
			for (int a = 0; a < 6; ++a) cnt[a] += __popcll(__ballot(sym[u] == (uint32_t)a));
		const int64_t rem = ntot - p0;
		const int nv = rem >= RB3_WIN ? RB3_WIN : rem > 0 ? (int)rem : 0;
		const uint32_t first = wave_read(sym[0], 0);
		uint32_t last = 7;
		if (nv > 0) {
			const int lu = (nv - 1) >> 6, ll = (nv - 1) & 63;
			const uint32_t v = lu == 0 ? sym[0] : lu == 1 ? sym[1] : lu == 2 ? sym[2] : sym[3];
			last = __shfl(v, ll);
		}
		if (lane == lw) my_nruns = nruns, my_first = first, my_last = last;
	}
	// partition the windows into slots: the largest aligned power-of-two groups with <= 48 runs
	const uint32_t prev_last = wave_up1(my_last);
	const int b = (lane > 0 && lane < nvw && my_nruns > 0 && prev_last == my_first) ? 1 : 0;
	const int e = lane < nvw ? my_nruns - b : 0;
	int P = e;
	P = (int)wave_incl_scan((uint32_t)P);
	int level = 0;
#pragma unroll
	for (int jl = 1; jl <= 5; ++jl) {
		const int sz = 1 << jl, a = lane & ~(sz - 1);
		const int Pa = __shfl(P, a), Pe = __shfl(P, (a + sz - 1) & 63), na = __shfl(my_nruns, a);
		const bool ok = (a + sz <= nvw) && (Pe - Pa + na <= RB3_RLE_CODES);
		if (ok && level == jl - 1) level = jl;
	}
	const bool start = lane < nvw && (lane & ((1 << level) - 1)) == 0;
	const uint32_t mask = (uint32_t)__ballot(start);
	if (lane < 8) {
		uint32_t v = lane == 0 ? cnt[0] : lane == 1 ? cnt[1] : lane == 2 ? cnt[2] : lane == 3 ? cnt[3] : lane == 4 ? cnt[4] :
			lane == 5 ? cnt[5] : lane == 6 ? (uint32_t)__popc(mask) : mask;
		gstat[g * 8 + lane] = v;
	}
}

/* pass 2: regenerate the symbols of each group and emit its slots and directory entry.
 * gpre[g*8 + 0..5] = symbol counts before the group, [6] = slots before the group. */
template<bool FROM_PLAIN>
__global__ void __launch_bounds__(64) k_pass2(IdxView old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot,
		const int64_t *jg, const uint32_t *gstat, const uint64_t *gpre, const uint64_t *tot, rb3_grp_t *grp, uint4 *slot16, int64_t ngrp, const unsigned long long *skip, int64_t abs_lim = RB3_ABS_LIMIT)
{
	if (RB3_REB_SKIP(skip)) return;
	__shared__ __attribute__((aligned(16))) uint8_t symbuf[RB3_WIN];
	__shared__ uint64_t ball[12];
	__shared__ uint32_t csym[RB3_RLE_CODES + 16], clen[RB3_RLE_CODES + 16];
	__shared__ uint32_t code16[RB3_RLE_CODES / 2];
	const int lane = threadIdx.x;
	const int64_t g = blockIdx.x;
	const int64_t W = (ntot >> RB3_WIN_BITS) + 1;
	const int nvw = (int)(W - g * RB3_GRP_WINS < RB3_GRP_WINS ? W - g * RB3_GRP_WINS : RB3_GRP_WINS);
	const uint32_t mask = gstat[g * 8 + 7];
	const uint64_t slot0 = gpre[g * 8 + 6];
	if (lane == 0) {
		rb3_grp_t e;
		uint64_t c = 0; // C[a] of the merged BWT from the symbol totals of the scan
		for (int a = 0; a < 6; ++a) { e.cnt[a] = c + gpre[g * 8 + a]; c += tot[a]; }
		e.slot0 = (uint32_t)slot0, e.mask = mask, e.spare = 0;
		grp[g] = e;
	}
	int64_t j = FROM_PLAIN ? 0 : jg[g];
	uint32_t cnt[6] = {0, 0, 0, 0, 0, 0}, rel[6] = {0, 0, 0, 0, 0, 0};
	int64_t sidx = (int64_t)slot0 - 1;
	int slot_w0 = 0, slot_sz = 1, nc = 0;
	uint32_t abs_base = 0; // lanes 1..6: the LF base of the group start if the headers are absolute (RB3_ABS_HEADERS)
	if (RB3_ABS_HEADERS(ntot, abs_lim) && lane >= 1 && lane <= 6) {
		uint64_t cb = gpre[g * 8 + lane - 1];
		for (int a = 0; a < lane - 1; ++a) cb += tot[a];
		abs_base = (uint32_t)cb;
	}
	for (int lw = 0; lw < nvw; ++lw) {
		const int64_t p0 = (g * RB3_GRP_WINS + lw) << RB3_WIN_BITS;
		uint32_t sym[4];
		gen_window<FROM_PLAIN>(old, pos, b2, n2, ntot, p0, j, symbuf, sym, lane);
		if (mask >> lw & 1u) { // a new slot begins
			++sidx, slot_w0 = lw, nc = 0;
			const uint32_t above = lw == 31 ? 0u : mask >> (lw + 1);
			const int nxt = above ? lw + 1 + (__ffs(above) - 1) : nvw;
			slot_sz = nxt - lw;
#pragma unroll
			for (int a = 0; a < 6; ++a) rel[a] = cnt[a];
		}
		const int64_t srem = ntot - ((g * RB3_GRP_WINS + slot_w0) << RB3_WIN_BITS);
		const uint32_t nsym = srem <= 0 ? 0u : srem < (int64_t)slot_sz * RB3_WIN ? (uint32_t)srem : (uint32_t)(slot_sz * RB3_WIN);
		uint32_t hq = lane == 0 ? (uint32_t)(slot_w0 * RB3_WIN) | (slot_sz > 1 ? RB3_SLOT_RLE : 0u) :
			lane == 1 ? rel[0] : lane == 2 ? rel[1] : lane == 3 ? rel[2] : lane == 4 ? rel[3] : lane == 5 ? rel[4] :
			lane == 6 ? rel[5] : nsym;
		hq += abs_base;
		if (slot_sz == 1) { // bit-plane slot
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int p = 0; p < 3; ++p) {
					const uint64_t m = __ballot((sym[u] >> p) & 1u);
					if (lane == u * 3 + p) ball[u * 3 + p] = m;
				}
			__syncthreads();
			if (lane < 8) {
				const uint32_t *b32 = (const uint32_t*)ball;
				uint4 v;
				v.x = hq;
				v.y = b32[(lane >> 1) * 6 + 0 + (lane & 1)];
				v.z = b32[(lane >> 1) * 6 + 2 + (lane & 1)];
				v.w = b32[(lane >> 1) * 6 + 4 + (lane & 1)];
				slot16[sidx * 8 + lane] = v;
			}
			__syncthreads();
		} else { // run slot: append this window's runs, merging across the window boundary
			uint64_t H[4];
			window_heads(sym, lane, H);
			const int64_t rem = ntot - p0;
			const int nv = rem >= RB3_WIN ? RB3_WIN : rem > 0 ? (int)rem : 0;
			const int nr = __popcll(H[0]) + __popcll(H[1]) + __popcll(H[2]) + __popcll(H[3]);
			const uint32_t fs = wave_read(sym[0], 0);
			const int mg = (nc > 0 && nr > 0 && csym[nc - 1] == fs) ? 1 : 0;
			int hb = 0;
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				if (H[u] >> lane & 1ull) {
					const int r = hb + __popcll(H[u] & ((1ull << lane) - 1ull));
					const uint64_t up = lane == 63 ? 0ull : H[u] >> (lane + 1) << (lane + 1);
					int nxt = nv;
					if (up) nxt = 64 * u + (__ffsll((unsigned long long)up) - 1);
					else {
						for (int v = u + 1; v < 4; ++v)
							if (H[v]) { nxt = 64 * v + (__ffsll((unsigned long long)H[v]) - 1); break; }
					}
					const int len = nxt - (64 * u + lane);
					const int idx = nc + r - mg;
					if (r == 0 && mg) clen[idx] += (uint32_t)len;
					else csym[idx] = sym[u], clen[idx] = (uint32_t)len;
				}
				hb += __popcll(H[u]);
			}
			nc += nr - mg;
			__syncthreads();
			if (lw == slot_w0 + slot_sz - 1) { // flush the slot
				{ // cumulative ends; the unused codes repeat the slot's last end
					const uint32_t cend = wave_incl_scan(lane < nc ? clen[lane] : 0u);
					if (lane < RB3_RLE_CODES) ((uint16_t*)code16)[lane] = (uint16_t)RB3_RUN_CODE(cend, lane < nc ? csym[lane] : 7u);
				}
				__syncthreads();
				if (lane < 8) {
					uint4 v;
					v.x = hq, v.y = code16[lane * 3], v.z = code16[lane * 3 + 1], v.w = code16[lane * 3 + 2];
					slot16[sidx * 8 + lane] = v;
				}
				__syncthreads();
			}
		}
#pragma unroll
		for (int u = 0; u < 4; ++u)
#pragma unroll
			for (int a = 0; a < 6; ++a) cnt[a] += __popcll(__ballot(sym[u] == (uint32_t)a));
	}
}

/* ----------------------------------------------------------------------------------------- */
/* window-parallel rebuild (same result as k_pass1/k_pass2, one wave per 256-symbol window)     */
/* ----------------------------------------------------------------------------------------- */

/* jw[w] = #{rows kb : pos[kb] < w * 256}, w = 0..nwin */
__global__ void __launch_bounds__(256) k_win_rows(const int64_t *pos, int64_t n2, int64_t *jw, int64_t nwin, const unsigned long long *skip)
{
	if (RB3_REB_SKIP(skip)) return;
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n2) return;
	int64_t a = i == 0 ? -1 : pos[i - 1] >> RB3_WIN_BITS;
	int64_t b = i == n2 ? nwin : pos[i] >> RB3_WIN_BITS;
	if (a < -1) a = -1;
	if (b > nwin) b = nwin;
	for (int64_t w = a + 1; w <= b; ++w) jw[w] = i;
}

/* Run-space short cut of k_pass1w for the common window of a compressible index: its old symbols lie inside ONE
 * run slot and at most three batch rows land in it.  Then the window is (at most 48 clipped runs) + (<= 3 single
 * symbols), and its statistics and run list follow from those ~50 items without ever touching 256 symbols:
 * lane q clips run q to the window, splits it at the rows that fall inside, the items are compacted in order,
 * adjacent items with equal symbols are merged.  Returns false (nothing written) if the window does not qualify;
 * the caller then takes the symbol path.  The planes are NOT produced (wstat gets RB3_WSTAT_NOPLANES): a window
 * like this almost always ends up in a run slot, and k_pass2w rebuilds the planes from the run list otherwise. */
#define RB3_WSTAT_NOPLANES 0x8000u
/* MR = most batch rows per window the short cut takes: every lane walks all MR rows, so a larger MR costs every window
 * (measured: 3 is best from ~85 indexed relatives up, 7 at 40, where a window receives 6 rows on average) */

template<int MR>
__device__ __forceinline__ bool window_runs_fast(const IdxView &old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot, int64_t w, int64_t j,
		int lane, uint32_t *sh /* >= 160 words of LDS of this wave */, uint4 *wstat, uint16_t *wruns, int64_t ws /* the window's place in the scratch arrays */)
{
	const int64_t p0 = w << RB3_WIN_BITS;
	if (ntot - p0 < RB3_WIN) return false; // the last window
	// the rows that land in this window (at most MR, else the symbol path)
	int64_t r = INT64_MAX;
	uint32_t rs = 7;
	if (lane <= MR && j + lane < n2) r = pos[j + lane];
	const bool in = r < p0 + RB3_WIN;
	const uint32_t inm = (uint32_t)__ballot(in);
	if (inm >> MR) return false;
	const int nb2 = __popc(inm);
	if (in) { if (r < p0) r = p0; rs = b2[j + lane]; }
	const int64_t a1 = p0 - j;
	const int nold = RB3_WIN - nb2;
	if (a1 < 0 || a1 + nold > old.n) return false;
	// one run slot must hold the whole old range
	const int64_t wa = a1 >> RB3_WIN_BITS, wb = (a1 + nold - 1) >> RB3_WIN_BITS;
	const int64_t ga = wa >> (RB3_GRP_BITS - RB3_WIN_BITS), gb = wb >> (RB3_GRP_BITS - RB3_WIN_BITS);
	if (ga != gb) return false;
	const uint64_t sm = old.grp64[ga * 8 + 6];
	const uint32_t sa = (uint32_t)sm + __popc((uint32_t)(sm >> 32) & ((2u << ((uint32_t)wa & (RB3_GRP_WINS - 1))) - 1u)) - 1u;
	const uint32_t sb = (uint32_t)sm + __popc((uint32_t)(sm >> 32) & ((2u << ((uint32_t)wb & (RB3_GRP_WINS - 1))) - 1u)) - 1u;
	if (sa != sb) return false;
	const uint32_t *sp = (const uint32_t*)(old.slot16 + (int64_t)sa * 8);
	const uint32_t hdr0 = sp[0];
	if (!(hdr0 & RB3_SLOT_RLE)) return false;
	const int A = (int)(a1 - ((ga << RB3_GRP_BITS) + (hdr0 & 0xFFFFu))); // the window's old range starts at offset A of the slot
	// the rows: old-local offset o_i = (new-local position) - i and symbol, in every lane
	int o[MR];
	uint32_t sy[MR];
#pragma unroll
	for (int i = 0; i < MR; ++i) {
		o[i] = i < nb2 ? (int)((int64_t)((uint64_t)wave_read((uint32_t)r, i) | (uint64_t)wave_read((uint32_t)((uint64_t)r >> 32), i) << 32) - p0) - i : 0x7fffffff;
		sy[i] = wave_read(rs, i);
	}
	// lane q: run q of the slot, clipped to the window, in old-local coordinates
	uint32_t inc = 0, rsym = 7; // inc: where the run ends
	if (lane < RB3_RLE_CODES) {
		const uint32_t word = sp[(lane / 6) * 4 + 1 + (lane % 6) / 2];
		const uint32_t code = (lane & 1) ? word >> 16 : word & 0xFFFFu;
		rsym = code & 7u, inc = RB3_RUN_END(code);
	}
	uint32_t rst = wave_up1(inc); // ... and where it starts: the end of the code before it
	if (lane == 0) rst = 0u;
	if (lane >= RB3_RLE_CODES || rsym == 7u) rst = inc; // (unused codes: empty)
	int cs = (int)rst - A, ce = (int)inc - A;
	cs = cs < 0 ? 0 : cs, ce = ce > nold ? nold : ce;
	const bool valid = ce > cs;
	const uint64_t vm = __ballot(valid);
	if (vm == 0) return false;
	const bool lastp = valid && (vm >> lane) == 1ull; // the last piece also takes the rows behind the last old symbol
	// items of this lane, in order: [part of the run] row [part] row ... [rest]; parts may be empty.  Item slot 2i is the
	// part before row i, slot 2i+1 row i, the last slot the rest; `present` says which exist.
	uint32_t present = 0, plen[MR + 1];
	bool mine[MR];
	{
		int at = cs;
#pragma unroll
		for (int i = 0; i < MR; ++i) {
			mine[i] = valid && i < nb2 && o[i] >= cs && (o[i] < ce || (lastp && o[i] <= ce));
			plen[i] = mine[i] && o[i] > at ? (uint32_t)(o[i] - at) : 0u;
			if (plen[i]) present |= 1u << (2 * i);
			if (mine[i]) present |= 1u << (2 * i + 1), at = o[i];
		}
		plen[MR] = valid && ce > at ? (uint32_t)(ce - at) : 0u;
		if (plen[MR]) present |= 1u << (2 * MR);
	}
	uint32_t ic = __popc(present), ioff = ic;
	ioff = wave_incl_scan((uint32_t)ioff);
	const int nitems = (int)wave_read(ioff, 63);
	if (nitems > 64) return false;
	ioff -= ic;
	uint32_t *it = sh; // items: sym | len << 8
#pragma unroll
	for (int i = 0; i <= MR; ++i) {
		if (present >> (2 * i) & 1u) it[ioff + __popc(present & ((1u << (2 * i)) - 1u))] = rsym | plen[i] << 8;
		if (i < MR && (present >> (2 * i + 1) & 1u)) it[ioff + __popc(present & ((1u << (2 * i + 1)) - 1u))] = sy[i] | 1u << 8;
	}
	if (lane < 8) sh[128 + lane] = 0u; // symbol counts
	wave_sync();
	// merge equal neighbours: one lane per item
	uint32_t me = lane < nitems ? it[lane] : 7u;
	const uint32_t msym = me & 0xFFu, mlen = lane < nitems ? me >> 8 : 0u;
	uint32_t prev = wave_up1(msym);
	if (lane == 0) prev = 8u;
	const bool head = lane < nitems && msym != prev;
	const uint64_t H = __ballot(head);
	const int nruns = __popcll(H);
	if (nruns > RB3_RLE_CODES) return false;
	uint32_t pinc = mlen;
	pinc = wave_incl_scan((uint32_t)pinc);
	const uint32_t mypos = pinc - mlen; // new-local position of this item
	const uint64_t above = lane == 63 ? 0ull : H >> (lane + 1);
	const int nh = above ? lane + 1 + (__ffsll((unsigned long long)above) - 1) : 64;
	uint32_t npos = __shfl(mypos, nh & 63);
	if (nh >= nitems) npos = RB3_WIN;
	if (head) {
		const uint32_t rl = npos - mypos;
		wruns[ws * RB3_RLE_CODES + __popcll(H & ((1ull << lane) - 1ull))] = (uint16_t)((rl - 1u) << 3 | msym);
		atomicAdd(&sh[128 + msym], rl);
	}
	wave_sync();
	if (lane == 0) {
		const uint32_t first = it[0] & 0xFFu, last = it[nitems - 1] & 0xFFu;
		uint4 v;
		v.x = sh[128] | sh[129] << 16, v.y = sh[130] | sh[131] << 16, v.z = sh[132] | sh[133] << 16;
		v.w = (uint32_t)nruns | RB3_WSTAT_NOPLANES | first << 16 | last << 24;
		wstat[ws] = v;
	}
	wave_sync();
	return true;
}

/* per window: symbols -> statistics (wstat: 6 x u16 counts, first, last, u16 runs = 16 B) and the
 * three bit planes (wplane: 24 dwords, the payload of a bit-plane slot) */
/* LISTED: only the windows of the groups glist[0 .. *nglist) (the groups the run-space rebuild k_reb_group left over),
 * with a grid-stride loop: the length of the list is only known on the device.  The scratch arrays (wstat, wplane, wruns)
 * are then indexed by the place in the list, 32 windows per entry, and hold lcap entries: a longer list raises *lover and
 * nothing is written (the host redoes the rebuild with full-size scratch). */
template<bool FROM_PLAIN, int MR = 3, bool LISTED = false>
__global__ void __launch_bounds__(64 * RB3_REB_WAVES) k_pass1w(IdxView old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot,
		const int64_t *jw, uint4 *wstat, uint32_t *wplane, uint16_t *wruns, int64_t nwin, const unsigned long long *skip,
		const uint32_t *glist = nullptr, const uint32_t *nglist = nullptr, uint32_t lcap = 0)
{
	__shared__ __attribute__((aligned(16))) uint8_t symbuf_[RB3_REB_WAVES][RB3_WIN];
	if (RB3_REB_SKIP(skip)) return;
	if (LISTED && *nglist > lcap) return;
	__shared__ uint64_t ball_[RB3_REB_WAVES][12];
	__shared__ uint32_t fast_[RB3_REB_WAVES][160];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint8_t *symbuf = symbuf_[wave];
	uint64_t *ball = ball_[wave];
	const int64_t nunits = LISTED ? (int64_t)*nglist * RB3_GRP_WINS : nwin;
	for (int64_t v = (int64_t)blockIdx.x * RB3_REB_WAVES + wave; v < nunits; v += LISTED ? (int64_t)gridDim.x * RB3_REB_WAVES : nunits) {
	const int64_t w = LISTED ? (int64_t)glist[v >> 5] * RB3_GRP_WINS + (v & (RB3_GRP_WINS - 1)) : v;
	const int64_t ws = v; // == w unless LISTED
	if (w >= nwin) continue;
	const int64_t p0 = w << RB3_WIN_BITS;
	int64_t j = FROM_PLAIN ? 0 : jw[w];
	if (!FROM_PLAIN && old.dense == 0 && window_runs_fast<MR>(old, pos, b2, n2, ntot, w, j, lane, fast_[wave], wstat, wruns, ws)) continue;
	uint32_t sym[4];
	gen_window<FROM_PLAIN>(old, pos, b2, n2, ntot, p0, j, symbuf, sym, lane);
	uint64_t H[4];
	window_heads(sym, lane, H);
	const uint32_t nruns = __popcll(H[0]) + __popcll(H[1]) + __popcll(H[2]) + __popcll(H[3]);
	uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
	for (int u = 0; u < 4; ++u) {
#pragma unroll
		for (int a = 0; a < 6; ++a) cnt[a] += __popcll(__ballot(sym[u] == (uint32_t)a));
#pragma unroll
		for (int p = 0; p < 3; ++p) {
			const uint64_t m = __ballot((sym[u] >> p) & 1u);
			if (lane == u * 3 + p) ball[u * 3 + p] = m;
		}
	}
	const int64_t rem = ntot - p0;
	const int nv = rem >= RB3_WIN ? RB3_WIN : rem > 0 ? (int)rem : 0;
	const uint32_t first = wave_read(sym[0], 0);
	uint32_t last = 7;
	if (nv > 0) {
		const int lu = (nv - 1) >> 6, ll = (nv - 1) & 63;
		const uint32_t v = lu == 0 ? sym[0] : lu == 1 ? sym[1] : lu == 2 ? sym[2] : sym[3];
		last = __shfl(v, ll);
	}
	wave_sync();
	if (lane < 24) wplane[ws * 24 + lane] = ((const uint32_t*)ball)[lane];
	if (lane == 0) {
		uint4 v;
		v.x = cnt[0] | cnt[1] << 16, v.y = cnt[2] | cnt[3] << 16, v.z = cnt[4] | cnt[5] << 16;
		v.w = nruns | first << 16 | last << 24;
		wstat[ws] = v;
	}
	// the runs as codes (len-1) << 3 | sym: what a run slot is assembled from (a window with more than 48
	// runs can only become a bit-plane slot: nothing to prepare -- most windows of an index of reads)
	int hb = 0;
	if (nruns <= RB3_RLE_CODES) {
#pragma unroll
	for (int u = 0; u < 4; ++u) {
		if (H[u] >> lane & 1ull) {
			const int r = hb + __popcll(H[u] & ((1ull << lane) - 1ull));
			if (r < RB3_RLE_CODES) {
				const uint64_t up = lane == 63 ? 0ull : H[u] >> (lane + 1) << (lane + 1);
				int nxt = nv;
				if (up) nxt = 64 * u + (__ffsll((unsigned long long)up) - 1);
				else {
					for (int v = u + 1; v < 4; ++v)
						if (H[v]) { nxt = 64 * v + (__ffsll((unsigned long long)H[v]) - 1); break; }
				}
				const int len = nxt - (64 * u + lane);
				wruns[ws * RB3_RLE_CODES + r] = (uint16_t)((uint32_t)(len - 1) << 3 | sym[u]);
			}
		}
		hb += __popcll(H[u]);
	}
	}
	wave_sync();
	} // windows of this wave
}

/* per group of 32 windows: the slot partition (largest aligned power-of-two window groups with
 * <= 48 runs) and the group's symbol counts; same output as k_pass1 */
template<bool LISTED = false>
__global__ void __launch_bounds__(64) k_decide(const uint4 *wstat, int64_t ntot, uint32_t *gstat, int64_t ngrp, const unsigned long long *skip,
		const uint32_t *glist = nullptr, const uint32_t *nglist = nullptr, uint32_t lcap = 0, unsigned long long *lover = nullptr)
{
	if (RB3_REB_SKIP(skip)) return;
	if (LISTED && *nglist > lcap) { // the list does not fit the scratch arrays: tell the host (the slot counts of these groups stay undefined)
		if (blockIdx.x == 0 && threadIdx.x == 0) *lover = 1;
		return;
	}
	const int lane = threadIdx.x;
	const int64_t nunits = LISTED ? (int64_t)*nglist : ngrp;
	for (int64_t gu = blockIdx.x; gu < nunits; gu += LISTED ? (int64_t)gridDim.x : nunits) {
	const int64_t g = LISTED ? (int64_t)glist[gu] : gu;
	const int64_t W = (ntot >> RB3_WIN_BITS) + 1;
	const int nvw = (int)(W - g * RB3_GRP_WINS < RB3_GRP_WINS ? W - g * RB3_GRP_WINS : RB3_GRP_WINS);
	uint4 st = make_uint4(0, 0, 0, 7u << 16 | 7u << 24);
	if (lane < nvw) st = wstat[gu * RB3_GRP_WINS + lane]; // (gu == g unless LISTED)
	const int my_nruns = (int)(st.w & 0x7FFFu); // (bit 15: RB3_WSTAT_NOPLANES)
	const uint32_t my_first = st.w >> 16 & 0xFFu, my_last = st.w >> 24;
	uint32_t cnt[6] = { st.x & 0xFFFFu, st.x >> 16, st.y & 0xFFFFu, st.y >> 16, st.z & 0xFFFFu, st.z >> 16 };
#pragma unroll
	for (int a = 0; a < 6; ++a)
		cnt[a] = wave_read(wave_incl_scan(cnt[a]), 63);
	const uint32_t prev_last = wave_up1(my_last);
	const int b = (lane > 0 && lane < nvw && my_nruns > 0 && prev_last == my_first) ? 1 : 0;
	const int e = lane < nvw ? my_nruns - b : 0;
	int P = e;
	P = (int)wave_incl_scan((uint32_t)P);
	int level = 0;
#pragma unroll
	for (int jl = 1; jl <= 5; ++jl) {
		const int sz = 1 << jl, a = lane & ~(sz - 1);
		const int Pa = __shfl(P, a), Pe = __shfl(P, (a + sz - 1) & 63), na = __shfl(my_nruns, a);
		const bool ok = (a + sz <= nvw) && (Pe - Pa + na <= RB3_RLE_CODES);
		if (ok && level == jl - 1) level = jl;
	}
	const bool start = lane < nvw && (lane & ((1 << level) - 1)) == 0;
	const uint32_t mask = (uint32_t)__ballot(start);
	if (lane < 8) {
		uint32_t v = lane == 0 ? cnt[0] : lane == 1 ? cnt[1] : lane == 2 ? cnt[2] : lane == 3 ? cnt[3] : lane == 4 ? cnt[4] :
			lane == 5 ? cnt[5] : lane == 6 ? (uint32_t)__popc(mask) : mask;
		gstat[g * 8 + lane] = v;
	}
	} // groups of this block
}

/* per window: emit its slot (bit-plane slots straight from the cached planes; the wave of the first
 * window of a run slot gathers the slot's <= 48 codes from the run lists of its windows, one lane per
 * code, merging runs that continue across window boundaries); window 0 of a group also writes the
 * directory entry */
template<bool LISTED = false>
__global__ void __launch_bounds__(64 * RB3_REB_WAVES) k_pass2w(const uint4 *wstat, const uint32_t *wplane, const uint16_t *wruns, int64_t ntot, const uint32_t *gstat,
		const uint64_t *gpre, const uint64_t *tot, rb3_grp_t *grp, uint4 *slot16, int64_t nwin, const unsigned long long *skip,
		const uint32_t *glist = nullptr, const uint32_t *nglist = nullptr, uint32_t lcap = 0, uint64_t slot_cap = ~0ull,
		int64_t all_nwin = -1, int64_t all_ntot = -1, int64_t abs_lim = RB3_ABS_LIMIT) // (a chunk of a larger build: windows / symbols of the WHOLE index, which decide the header kind)
{
	if (RB3_REB_SKIP(skip)) return;
	if (LISTED && *nglist > lcap) return;
	if (tot[6] > slot_cap) return; // the slot array was sized by an estimate: the host looks at the total and emits again
	if (all_nwin < 0) all_nwin = nwin, all_ntot = ntot;
#ifdef RB3_ABL
	if (LISTED) return;
#endif
	__shared__ uint32_t sP_[RB3_REB_WAVES][RB3_GRP_WINS + 1], sB_[RB3_REB_WAVES][RB3_GRP_WINS], sNr_[RB3_REB_WAVES][RB3_GRP_WINS];
	__shared__ uint32_t code16_[RB3_REB_WAVES][RB3_RLE_CODES / 2];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t *sP = sP_[wave], *sB = sB_[wave], *sNr = sNr_[wave], *code16 = code16_[wave];
	const int64_t ngrp_all = (nwin + RB3_GRP_WINS - 1) / RB3_GRP_WINS;
	const int64_t nunits = LISTED ? (int64_t)*nglist : ngrp_all;
	for (int64_t gu = blockIdx.x; gu < nunits; gu += LISTED ? (int64_t)gridDim.x : nunits) {
	const int64_t g = LISTED ? (int64_t)glist[gu] : gu; // one block per group at a time, each wave takes a quarter of its windows
	const int nvw = (int)(nwin - g * RB3_GRP_WINS < RB3_GRP_WINS ? nwin - g * RB3_GRP_WINS : RB3_GRP_WINS);
	const uint32_t mask = gstat[g * 8 + 7];
	const uint64_t slot0 = gpre[g * 8 + 6];
	if (threadIdx.x == 0) {
		rb3_grp_t e;
		uint64_t c = 0;
		for (int a = 0; a < 6; ++a) { e.cnt[a] = c + gpre[g * 8 + a]; c += tot[a]; }
		e.slot0 = (uint32_t)slot0, e.mask = mask, e.spare = 0;
		grp[g] = e;
	}
	uint32_t abs_base = 0; // lanes 1..6: the LF base of the group start if the headers are absolute (RB3_ABS_HEADERS)
	if (RB3_ABS_HEADERS(all_ntot, abs_lim) && lane >= 1 && lane <= 6) {
		uint64_t cb = gpre[g * 8 + lane - 1];
		for (int a = 0; a < lane - 1; ++a) cb += tot[a];
		abs_base = (uint32_t)cb;
	}
	// Every window its own slot: all of them bit-plane slots (a run slot has at least two windows) -- a group of an index of
	// reads.  A wave then writes its eight slots with ONE store: lane l = slice l & 7 of window (l >> 3).  The symbol counts of the
	// windows before (header words 1..6) come from a wave scan over the 32 window records, two 16-bit fields per dword (the counts
	// of a group stay below 2^14).  Windows of the run-space short cut (no planes cached) take the general path.
	const uint32_t fullmask = nvw >= 32 ? 0xFFFFFFFFu : ((1u << nvw) - 1u);
	if (mask == fullmask) {
		uint4 st = make_uint4(0, 0, 0, 0);
		if (lane < nvw) st = wstat[gu * RB3_GRP_WINS + lane];
		if (__ballot(lane < nvw && (st.w & RB3_WSTAT_NOPLANES)) == 0ull) {
			const uint32_t e0 = wave_incl_scan(st.x) - st.x, e1 = wave_incl_scan(st.y) - st.y, e2 = wave_incl_scan(st.z) - st.z; // (lanes >= nvw add 0)
			const int lw = wave * (RB3_GRP_WINS / RB3_REB_WAVES) + (lane >> 3), j = lane & 7;
			const uint32_t p0 = (uint32_t)__shfl((int)e0, lw), p1 = (uint32_t)__shfl((int)e1, lw), p2 = (uint32_t)__shfl((int)e2, lw);
			if (lw < nvw) {
				const int64_t w = g * RB3_GRP_WINS + lw, ws = gu * RB3_GRP_WINS + lw;
				const int64_t srem = ntot - (w << RB3_WIN_BITS);
				const uint32_t nsym = srem <= 0 ? 0u : srem < RB3_WIN ? (uint32_t)srem : (uint32_t)RB3_WIN;
				uint32_t hq = j == 0 ? (uint32_t)(lw * RB3_WIN) : j == 1 ? (p0 & 0xFFFFu) : j == 2 ? p0 >> 16 : j == 3 ? (p1 & 0xFFFFu) : j == 4 ? p1 >> 16 :
					j == 5 ? (p2 & 0xFFFFu) : j == 6 ? p2 >> 16 : nsym;
				hq += (uint32_t)__shfl((int)abs_base, j); // (abs_base lives in lanes 1..6 of the wave, 0 elsewhere)
				const uint32_t *pl = wplane + ws * 24;
				uint4 v;
				v.x = hq;
				v.y = pl[(j >> 1) * 6 + 0 + (j & 1)];
				v.z = pl[(j >> 1) * 6 + 2 + (j & 1)];
				v.w = pl[(j >> 1) * 6 + 4 + (j & 1)];
				slot16[((int64_t)slot0 + lw) * 8 + j] = v;
			}
			continue; // next group
		}
	}
	for (int lw = wave * (RB3_GRP_WINS / RB3_REB_WAVES); lw < (wave + 1) * (RB3_GRP_WINS / RB3_REB_WAVES) && lw < nvw; ++lw) {
		if (!(mask >> lw & 1u)) continue; // not the first window of a slot
		const int64_t w = g * RB3_GRP_WINS + lw;
		const int64_t ws = gu * RB3_GRP_WINS + lw; // the window in the scratch arrays (== w unless LISTED)
		const int64_t sidx = (int64_t)slot0 + __popc(mask & ((2u << lw) - 1u)) - 1;
		const uint32_t above = lw == 31 ? 0u : mask >> (lw + 1);
		const int slot_sz = (above ? lw + 1 + (__ffs(above) - 1) : nvw) - lw;
		// symbol counts of the group's windows before this slot
		uint4 st = make_uint4(0, 0, 0, 0);
		if (lane < lw) st = wstat[gu * RB3_GRP_WINS + lane];
		uint64_t s0 = (uint64_t)(st.x & 0xFFFFu) | (uint64_t)(st.x >> 16) << 20 | (uint64_t)(st.y & 0xFFFFu) << 40;
		uint64_t s1 = (uint64_t)(st.y >> 16) | (uint64_t)(st.z & 0xFFFFu) << 20 | (uint64_t)(st.z >> 16) << 40;
		for (int d = 16; d >= 1; d >>= 1) s0 += __shfl_xor(s0, d), s1 += __shfl_xor(s1, d);
		const uint32_t rel[6] = { (uint32_t)(s0 & 0xFFFFF), (uint32_t)(s0 >> 20 & 0xFFFFF), (uint32_t)(s0 >> 40),
		                          (uint32_t)(s1 & 0xFFFFF), (uint32_t)(s1 >> 20 & 0xFFFFF), (uint32_t)(s1 >> 40) };
		const int64_t srem = ntot - (w << RB3_WIN_BITS);
		const uint32_t nsym = srem <= 0 ? 0u : srem < (int64_t)slot_sz * RB3_WIN ? (uint32_t)srem : (uint32_t)(slot_sz * RB3_WIN);
		uint32_t hq = lane == 0 ? (uint32_t)(lw * RB3_WIN) | (slot_sz > 1 ? RB3_SLOT_RLE : 0u) :
			lane == 1 ? rel[0] : lane == 2 ? rel[1] : lane == 3 ? rel[2] : lane == 4 ? rel[3] : lane == 5 ? rel[4] :
			lane == 6 ? rel[5] : nsym;
		hq += abs_base;
		if (slot_sz == 1) { // bit-plane slot: header + the cached planes
			const uint32_t wflags = wstat[ws].w;
			if (wflags & RB3_WSTAT_NOPLANES) { // k_pass1w took the run-space short cut: make the planes from the run list (<= 48 runs)
				uint32_t len = 0, sy = 7;
				if (lane < (int)(wflags & 0x7FFFu)) { const uint32_t c = wruns[ws * RB3_RLE_CODES + lane]; sy = c & 7u, len = (c >> 3) + 1u; }
				uint32_t inc = len;
				inc = wave_incl_scan((uint32_t)inc);
				uint16_t *rst = (uint16_t*)code16; // 48 run starts
				uint8_t *rsy = (uint8_t*)sB;       // 48 run symbols
				uint32_t *pw = sNr;                // 24 plane words
				if (lane < RB3_RLE_CODES) rst[lane] = (uint16_t)(len ? inc - len : 0xFFFFu), rsy[lane] = (uint8_t)sy;
				wave_sync();
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t off = 64 * u + lane;
					int q = 0;
#pragma unroll
					for (int d = 32; d >= 1; d >>= 1)
						if (q + d < RB3_RLE_CODES && rst[q + d] <= off) q += d;
					const uint32_t s1 = off < nsym ? rsy[q] : 7u;
#pragma unroll
					for (int p = 0; p < 3; ++p) {
						const uint64_t m = __ballot((s1 >> p) & 1u);
						if (lane == u * 3 + p) pw[(u * 3 + p) * 2] = (uint32_t)m, pw[(u * 3 + p) * 2 + 1] = (uint32_t)(m >> 32);
					}
				}
				wave_sync();
				if (lane < 8) { // same word order as wplane
					uint4 v;
					v.x = hq;
					v.y = pw[(lane >> 1) * 6 + 0 + (lane & 1)];
					v.z = pw[(lane >> 1) * 6 + 2 + (lane & 1)];
					v.w = pw[(lane >> 1) * 6 + 4 + (lane & 1)];
					slot16[sidx * 8 + lane] = v;
				}
				wave_sync();
				continue;
			}
			if (lane < 8) {
				const uint32_t *pl = wplane + ws * 24;
				uint4 v;
				v.x = hq;
				v.y = pl[(lane >> 1) * 6 + 0 + (lane & 1)];
				v.z = pl[(lane >> 1) * 6 + 2 + (lane & 1)];
				v.w = pl[(lane >> 1) * 6 + 4 + (lane & 1)];
				slot16[sidx * 8 + lane] = v;
			}
			continue;
		}
		// run slot.  Lane k < slot_sz looks at window k: it contributes its runs minus the first one if that
		// continues the last run of the window before (same test as k_decide).
		uint4 sw = make_uint4(0, 0, 0, 7u << 16 | 7u << 24);
		if (lane < slot_sz) sw = wstat[ws + lane];
		const uint32_t nr = sw.w & 0x7FFFu, wfirst = sw.w >> 16 & 0xFFu, wlast = sw.w >> 24;
		const uint32_t prev_last = wave_up1(wlast);
		const uint32_t bm = (lane > 0 && lane < slot_sz && nr > 0 && prev_last == wfirst) ? 1u : 0u;
		const uint32_t e = lane < slot_sz ? nr - bm : 0u;
		uint32_t inc = e;
		inc = wave_incl_scan((uint32_t)inc);
		const int nc = (int)wave_read(inc, 63);
		if (lane < RB3_GRP_WINS) sP[lane] = lane < slot_sz ? inc - e : 0xFFFFu, sB[lane] = bm, sNr[lane] = nr;
		if (lane == 0) sP[RB3_GRP_WINS] = 0xFFFFu;
		wave_sync();
		{ // (wruns: per-window LENGTH codes (len - 1) << 3 | sym, an intermediate of this path; the slot gets cumulative ends)
			uint32_t len = 0u, csy = 7u;
			if (lane < nc) {
				int k = 0; // the last window whose first code is at or before this lane's
#pragma unroll
				for (int d = 16; d >= 1; d >>= 1)
					if (k + d < slot_sz && sP[k + d] <= (uint32_t)lane) k += d;
				const uint32_t r = (uint32_t)lane - sP[k] + sB[k];
				const uint32_t c0 = wruns[(ws + k) * RB3_RLE_CODES + r];
				len = (c0 >> 3) + 1u, csy = c0 & 7u;
				if (r == sNr[k] - 1u) // the window's last run may go on through the following windows
					for (int kk = k + 1; kk < slot_sz && sB[kk]; ++kk) {
						len += ((uint32_t)wruns[(ws + kk) * RB3_RLE_CODES] >> 3) + 1u;
						if (sNr[kk] > 1u) break;
					}
			}
			const uint32_t cend = wave_incl_scan(len); // (lanes behind the last run repeat the slot's last end)
			if (lane < RB3_RLE_CODES) ((uint16_t*)code16)[lane] = (uint16_t)RB3_RUN_CODE(cend, csy);
		}
		wave_sync();
		if (lane < 8) {
			uint4 v;
			v.x = hq, v.y = code16[lane * 3], v.z = code16[lane * 3 + 1], v.w = code16[lane * 3 + 2];
			slot16[sidx * 8 + lane] = v;
		}
		wave_sync();
	}
	} // groups of this block
}

/* ----------------------------------------------------------------------------------------- */
/* run-space rebuild: one wave per 8192-symbol GROUP of the new index                          */
/* ----------------------------------------------------------------------------------------- */

/* The per-window kernels above regenerate 256 symbols per window whatever the index looks like.  In a compressible index
 * (the target workload: many genomes of one species; runs of ~100 symbols) a group of 8192 output symbols is a few dozen
 * old runs plus the batch rows that land in it, and the whole rebuild of the group -- interleave (worker_mgins +
 * rope_insert_run, fm-index.c:237-249, rope.c:114-148), slot partition, run codes, header counts -- can be done on those
 * items without ever expanding a symbol.  One wave per group:
 *
 *   rows     the nb batch rows j0.. of the group: old offset k_r (non-decreasing), new offset q_r = k_r + r.  A row that lands strictly
 *            inside an old run of its OWN symbol -- or behind such rows at the same insertion point -- only lengthens that run: it is
 *            ABSORBED, i.e. it counts as an offset for everything behind it but produces no item (the usual fate of a genome
 *            merged into its relatives; the rows are streamed 64 per pass with a running maximum deciding who is absorbed)
 *   runs     i = 0..nR-1: the old runs that intersect the group's old range [A0, A0 + 8192 - nb), clipped, start S_i
 *   items    sorted by new offset; three kinds, and each knows its own index without a search through the other list:
 *            B_r (batch row r)           offset q_r,       index r + lb_r + C(r),  lb_r = #{i : S_i < k_r}: rank in a bitmap
 *                                                                                  of the run starts (one LDS round trip)
 *            C_r (the old run resumes)   offset q_r + 1,   index of B_r + 1        iff r is the last row with its k_r and
 *                                                                                  k_r lies strictly inside an old run
 *            A_i (start of old run i)    offset S_i + ub,  index i + ub + C',      ub = #{r : k_r <= S_i} = #{r : lb_r <= i},
 *                                                                                  C' = #{C_r : lb_r <= i}: prefix sums of a
 *                                                                                  histogram the rows fill by lb_r
 *   heads    items whose symbol differs from the item before: the maximal runs of the group
 *   slots    the same rule as k_decide (largest aligned power-of-two window blocks with <= 48 runs), evaluated from the
 *            number of heads before every window boundary (a histogram of the heads by window); all run slots of the group
 *            are cut out of the head list in one pass (one lane per code), single windows are expanded to bit planes.
 *
 * The slots go to a scratch array (at most RB3_RG_MAXSLOTS per group; final places need the scan over all groups) and
 * k_place copies them once the offsets are known.  A group that does not qualify -- a bit-plane slot in its old range,
 * too many rows / runs / slots for the tier, the last group -- is left to the next tier: k_reb_group with larger tables
 * (gkind[g] says which groups are still to do), then the per-window kernels (LISTED variants, from a list the last tier
 * appends to), so every index goes through; the host only starts the tiers that the average number of rows per group
 * makes worthwhile, and a fully bit-plane index (old.dense) never comes here at all.
 *
 * The global loads of a group form a chain jw -> (directory entries of the old range, batch rows) -> old slots; every
 * link is issued as one burst and jw one group ahead.  What is left is LDS latency: the kernel runs at ~2000 wave
 * instructions per group, waiting most of the time (profiles/r2_*). */
#define RB3_RG_MAXSLOTS 8
#define RB3_RG_WAVES 4
#define RB3_RG_FAILBUF 16

#ifdef RB3_PROF_REB
/* cycles per phase of reb_group_one, summed over groups (lane 0 of every wave): a kernel experiment, not in the release build */
__device__ unsigned long long g_reb_prof[16];
#define RB3_REB_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprof; tprof = t_; } while (0)
__device__ unsigned long long g_reb_why[8]; /* why groups were handed on: 0 rows, 1 slots of the old range, 2 bit-plane slot, 3 old runs, 4 row runs, 5 new slots, 6 last group */
#define RB3_REB_WHY(i) do { if (lane == 0) atomicAdd(&g_reb_why[i], 1ull); } while (0)
#else
#define RB3_REB_T(i) do {} while (0)
#define RB3_REB_WHY(i) do {} while (0)
#endif

template<int RMAX, int NBMAX>
struct RebLds {
	__attribute__((aligned(16))) uint32_t bits[260]; // bit s set <=> an old run starts at old-range offset s
	uint32_t it[RMAX + 2 * NBMAX + 2]; // items, then (in place) heads: offset << 3 | sym
	// The histogram of the batch rows is dead once the items are made, and what the slot phases accumulate into is only needed
	// from then on: one piece of LDS for both (a fifth of the structure; LDS is what bounds the waves per CU of this kernel,
	// and the kernel waits on LDS and global latency, so waves are what it needs).
	union {
		uint32_t PHC[RMAX + 2];        // histogram of the batch rows by lb_r, three packed fields: rows (bits 0-13), rows that became items (14-22), C items (23-31)
		struct {
			uint32_t stage[RB3_RG_MAXSLOTS * 24]; // payload of the group's slots: 48 run codes, or the 24 plane words
			uint32_t scnt[RB3_RG_MAXSLOTS * 8];   // symbol counts per slot
			uint32_t wh[34];               // heads per window
			uint32_t wexact;               // bit w: a head sits exactly on the start of window w
			uint16_t cH[34], hB[34];       // heads before / at-or-before every window boundary
			uint32_t slotA[RB3_RG_MAXSLOTS + 2];  // start window of every slot, slotA[nslots] = 32
			uint32_t sbase[RB3_RG_MAXSLOTS + 2];  // first code of every run slot in the group's code sequence
		};
	};
	uint16_t cum[260];             // set bits before word w of bits[]
	uint16_t S[RMAX + 2];          // old run starts (old-range offsets)
	uint8_t Ssym[RMAX + 2];
	uint32_t fail[RB3_RG_FAILBUF]; // groups this wave hands on to the window kernels, flushed with one atomic
};

/* number of heads h[0..n) whose offset (h >> 3) is < key */
__device__ __forceinline__ int lb_head(const uint32_t *h, int n, uint32_t key)
{
	int lo = 0;
	for (int step = n > 0 ? 1 << (31 - __clz(n)) : 0; step; step >>= 1)
		if (lo + step <= n && (h[lo + step - 1] >> 3) < key) lo += step;
	return lo;
}

/* one group; returns false (nothing of consequence written) if the group does not qualify for this tier.
 * j0, j1: jw[32 g], jw[32 g + 32], loaded by the caller one group ahead. */
template<int RMAX, int NBMAX>
__device__ __forceinline__ bool reb_group_one(const IdxView &old, const int64_t *pos, const uint8_t *b2, int64_t ntot, int64_t j0, int64_t j1,
		int64_t g, int lane, RebLds<RMAX, NBMAX> &L, uint32_t *gstat, uint4 *gslots
#ifdef RB3_PROF_REB
		, unsigned long long *tacc
#endif
		)
{
	const int64_t P0 = g << RB3_GRP_BITS;
#ifdef RB3_PROF_REB
	unsigned long long tprof = __builtin_readcyclecounter();
#endif
	if (ntot - P0 < RB3_GRP) { RB3_REB_WHY(6); return false; } // the last (partial) group goes through the window kernels
	const int64_t nb64 = j1 - j0;
	if (nb64 < 0 || nb64 > RB3_GRP - 64) { RB3_REB_WHY(0); return false; } // (a group that is nearly all new rows: the window kernels)
	const int nb = (int)nb64, nold = RB3_GRP - nb;
	const int64_t A0 = P0 - j0, A1 = A0 + nold;
	if (A0 < 0 || A1 > old.n) return false; // (only with an invalid pos[])
	RB3_REB_T(0);
#if defined(RB3_ABL) && RB3_ABL == 1
	return true;
#endif
	// the first batch rows of the group, requested now (registers), used after the old runs; the rest in a loop behind them
	constexpr int NCH = 2;
	int64_t rpos[NCH];
	uint8_t rsym[NCH];
#pragma unroll
	for (int c = 0; c < NCH; ++c) {
		const int r = c * 64 + lane;
		rpos[c] = r < nb ? pos[j0 + r] : 0;
		rsym[c] = r < nb ? b2[j0 + r] : (uint8_t)0;
	}
	{ // clear what the phases below accumulate into
		uint4 *bz = (uint4*)L.bits;
		bz[lane] = make_uint4(0, 0, 0, 0);
		if (lane == 0) bz[64] = make_uint4(0, 0, 0, 0);
	}
	wave_sync();
	// ---- the old runs of [A0, A1) ----
	int nR = 0;
	if (nold > 0) {
		const int64_t ga = A0 >> RB3_GRP_BITS, gb = (A1 - 1) >> RB3_GRP_BITS;
		const uint64_t sma = old.gsm[ga], smb = old.gsm[gb]; // (the compact copy of the slot words: neighbouring groups share a line)
		const uint32_t wa = ((uint32_t)A0 & (RB3_GRP - 1)) >> RB3_WIN_BITS, wb = ((uint32_t)(A1 - 1) & (RB3_GRP - 1)) >> RB3_WIN_BITS;
		const int64_t fa = (int64_t)((uint32_t)sma + __popc((uint32_t)(sma >> 32) & ((2u << wa) - 1u)) - 1u);
		const int64_t la = (int64_t)((uint32_t)smb + __popc((uint32_t)(smb >> 32) & ((2u << wb) - 1u)) - 1u);
		const int64_t slot0b = (int64_t)(uint32_t)smb;
		const int ns = (int)(la - fa + 1);
		if (ns <= 0 || ns * RB3_RLE_CODES > RMAX) { RB3_REB_WHY(1); return false; }
		const int j = lane & 7;
		constexpr int NPASS = (RMAX / RB3_RLE_CODES + 7) / 8;
		uint4 slv[NPASS];
#pragma unroll
		for (int p = 0; p < NPASS; ++p) { // all slots of the range in one burst
			const int q = p * 8 + (lane >> 3);
			slv[p] = make_uint4(RB3_SLOT_RLE, 0x00070007u, 0x00070007u, 0x00070007u);
			if (q < ns) slv[p] = old.slot16[(fa + q) * 8 + j];
		}
#pragma unroll
		for (int p = 0; p < NPASS; ++p) {
			if (p * 8 >= ns) break;
			const int q = p * 8 + (lane >> 3);
			const bool valid = q < ns;
			const uint4 sl = slv[p];
			const uint32_t hdr0 = oct_bcast0(sl.x, j);
			if (__any(valid && !(hdr0 & RB3_SLOT_RLE))) { RB3_REB_WHY(2); return false; } // a bit-plane slot: this group is rebuilt from symbols
			const int64_t sgrp = (gb != ga && fa + q >= slot0b) ? gb : ga;
			const uint32_t e[6] = { sl.y & 0xFFFFu, sl.y >> 16, sl.z & 0xFFFFu, sl.z >> 16, sl.w & 0xFFFFu, sl.w >> 16 };
			// cumulative codes: run i covers [end of the code before it, its own end) of the slot -- the first one of a lane starts where the lane below ended
			const int sbase = (int)(((sgrp << RB3_GRP_BITS) + (int64_t)(hdr0 & 0xFFFFu)) - A0); // the slot's start in the group's old range: may be negative
			int rel = sbase + (int)(oct_prev_end(pk_ends(sl.w), j) >> 16);
			int cs[6];
			uint32_t nv = 0, vm = 0;
#pragma unroll
			for (int i = 0; i < 6; ++i) {
				const int eni = sbase + (int)RB3_RUN_END(e[i]);
				const int st = rel < 0 ? 0 : rel, en = eni > nold ? nold : eni;
				cs[i] = st;
				if (valid && (e[i] & 7u) != 7u && en > st) vm |= 1u << i, ++nv;
				rel = eni;
			}
			const uint32_t inc = wave_incl_scan(nv);
			int o = nR + (int)(inc - nv);
#pragma unroll
			for (int i = 0; i < 6; ++i)
				if (vm >> i & 1u) {
					L.S[o] = (uint16_t)cs[i], L.Ssym[o] = (uint8_t)(e[i] & 7u);
					atomicOr(&L.bits[cs[i] >> 5], 1u << (cs[i] & 31));
					++o;
				}
			nR += (int)wave_read(inc, 63);
		}
		if (nR <= 0 || nR > RMAX) { RB3_REB_WHY(3); return false; }
	}
	RB3_REB_T(1);
	// ---- the batch rows ----
	for (int c = 0; c * 64 <= nR; ++c)
		if (c * 64 + lane <= nR) L.PHC[c * 64 + lane] = 0u;
	wave_sync();
	{ // cum[w] = set bits before word w
		const uint4 bw = ((const uint4*)L.bits)[lane];
		const uint32_t c0 = __popc(bw.x), c1 = __popc(bw.y), c2 = __popc(bw.z), c3 = __popc(bw.w), t4 = c0 + c1 + c2 + c3;
		const uint32_t base = wave_incl_scan(t4) - t4;
		ushort4 cv;
		cv.x = (uint16_t)base, cv.y = (uint16_t)(base + c0), cv.z = (uint16_t)(base + c0 + c1), cv.w = (uint16_t)(base + c0 + c1 + c2);
		((ushort4*)L.cum)[lane] = cv;
		if (lane == 0) L.cum[256] = (uint16_t)nR;
	}
	wave_sync();
	RB3_REB_T(2);
#if defined(RB3_ABL) && RB3_ABL == 2
	return true;
#endif
	// A row that falls strictly inside an old run of its own symbol changes nothing but the length of that run: it is ABSORBED
	// (no item; it still counts for the offsets of everything behind it).  Only a row whose symbol differs from the run it
	// splits, or that sits between two runs, becomes an item -- and so does every later row at the same insertion point, and
	// the run resumes behind the last of them.  In an index of relatives nearly every row is absorbed, whatever the batch
	// size: the work of a group is its old runs plus the variants.  The rows are streamed (64 per pass), nothing is stored.
	int nB = 0, nC = 0;
	{
		uint32_t cbad = 0, cF = 0, pk = 0xFFFFFFFFu; // rows-with-an-item so far / that count at the start of the current insertion point / k of the row before
		for (int c = 0; c * 64 < nb; ++c) {
			const int r = c * 64 + lane;
			const bool valid = r < nb;
			int64_t pp = 0;
			uint32_t sy = 0;
			if (c < NCH) { pp = c == 0 ? rpos[0] : rpos[NCH - 1]; sy = c == 0 ? rsym[0] : rsym[NCH - 1]; }
			else if (valid) pp = pos[j0 + r], sy = b2[j0 + r];
			const int64_t pfirst_next = (c + 1) * 64 < nb ? pos[j0 + (c + 1) * 64] : 0; // (one address for the wave)
			const int64_t k64 = pp - P0 - r;
			const uint32_t k = valid ? (uint32_t)(k64 < 0 ? 0 : k64 > nold ? nold : k64) : 0xFFFFFFFEu; // (clamped: only an invalid pos[] is outside)
			uint32_t kprev = wave_up1(k), kn = wave_down1(k);
			if (lane == 0) kprev = pk;
			if (lane == 63) kn = (c + 1) * 64 < nb ? (uint32_t)(pfirst_next - P0 - (c + 1) * 64) : 0xFFFFFFFEu;
			const bool khead = valid && k != kprev, kend = valid && kn != k;
			const uint32_t kk = valid ? k : 0u;
			const uint32_t bwd = L.bits[kk >> 5];
			const int lb = (int)L.cum[kk >> 5] + __popc(bwd & ((1u << (kk & 31)) - 1u)); // run starts before k
			const bool inside = valid && !((bwd >> (kk & 31)) & 1u) && lb >= 1 && (int)kk < nold; // strictly inside old run lb - 1
			const uint32_t ci = inside ? (uint32_t)L.Ssym[lb - 1] : 0xFFu;
			const uint32_t bad = valid && sy != ci ? 1u : 0u;
			const uint32_t bincl = wave_incl_scan(bad) + cbad, bexcl = bincl - bad;
			uint32_t F = wave_incl_max(khead ? bexcl : 0u);
			F = F > cF ? F : cF;
			const bool item = valid && bincl > F;          // a differing row at this insertion point, at or before this one
			const bool cf = kend && inside && item;        // ... and the old run goes on behind the last of them
			const uint64_t balB = __ballot(item), balC = __ballot(cf);
			const uint64_t lt = (1ull << lane) - 1ull;
			if (item) {
				const int idx = nB + __popcll(balB & lt) + lb + nC + __popcll(balC & lt);
				L.it[idx] = (kk + (uint32_t)r) << 3 | sy;
				if (cf) L.it[idx + 1] = (kk + (uint32_t)r + 1u) << 3 | ci;
			}
			if (valid) atomicAdd(&L.PHC[lb], 1u | (item ? 1u << 14 : 0u) | (cf ? 1u << 23 : 0u));
			nB += __popcll(balB), nC += __popcll(balC);
			if (nB > NBMAX) { RB3_REB_WHY(4); return false; } // too many rows that differ for this tier
			cbad = wave_read(bincl, 63), cF = wave_read(F, 63), pk = wave_read(k, 63);
		}
	}
	wave_sync();
	RB3_REB_T(3);
	{
		uint32_t carry = 0;
		for (int c = 0; c * 64 < nR; ++c) {
			const int i = c * 64 + lane;
			const uint32_t v = i < nR ? L.PHC[i] : 0u;
			const uint32_t inc = wave_incl_scan(v) + carry; // all three fields: rows / rows with an item / C items with lb <= i (no field overflows: <= 8192, 511, 511)
			if (i < nR) {
				const uint32_t rows = inc & 0x3FFFu, runs = inc >> 14 & 0x1FFu, cs = inc >> 23;
				L.it[i + (int)runs + (int)cs] = ((uint32_t)L.S[i] + rows) << 3 | (uint32_t)L.Ssym[i];
			}
			carry = wave_read(inc, 63);
		}
	}
	const int nI = nR + nB + nC;
	wave_sync();
	{ // the histogram has been read: its place now holds what the slot phases accumulate into
		L.scnt[lane] = 0u;
		if (lane < 34) L.wh[lane] = 0u;
		if (lane == 0) L.wexact = 0u;
#pragma unroll
		for (int q = 0; q < 3; ++q) L.stage[q * 64 + lane] = 0x00070007u; // unused run codes
	}
	wave_sync();
	RB3_REB_T(4);
	// ---- heads (maximal runs), compacted in place; heads per window ----
	int nH = 0;
	{
		uint32_t carry = 8u;
		for (int c = 0; c * 64 < nI; ++c) {
			const int i = c * 64 + lane;
			const bool valid = i < nI;
			const uint32_t x = valid ? L.it[i] : 7u, sy = x & 7u;
			uint32_t up = wave_up1(sy);
			if (lane == 0) up = carry;
			const bool head = valid && sy != up;
			const uint64_t bal = __ballot(head);
			if (head) {
				L.it[nH + __popcll(bal & ((1ull << lane) - 1ull))] = x; // never ahead of what has been read
				atomicAdd(&L.wh[x >> (3 + RB3_WIN_BITS)], 1u);
				if (((x >> 3) & (RB3_WIN - 1)) == 0u) atomicOr(&L.wexact, 1u << (x >> (3 + RB3_WIN_BITS)));
			}
			carry = wave_read(sy, 63);
			nH += __popcll(bal);
		}
	}
	if (lane == 0) L.it[nH] = (uint32_t)RB3_GRP << 3 | 7u;
	wave_sync();
	const uint32_t *H = L.it;
	RB3_REB_T(5);
#if defined(RB3_ABL) && RB3_ABL == 3
	return true;
#endif
	// ---- slot partition: the rule of k_decide on the number of runs that intersect each aligned block of windows ----
	{
		const uint32_t v = lane < RB3_GRP_WINS ? L.wh[lane] : 0u;
		const uint32_t c0 = wave_incl_scan(v) - v;
		if (lane <= RB3_GRP_WINS) L.cH[lane] = (uint16_t)c0, L.hB[lane] = (uint16_t)(c0 + (lane < RB3_GRP_WINS ? (L.wexact >> lane) & 1u : 0u));
	}
	wave_sync();
	int level = 0;
	if (lane < RB3_GRP_WINS) {
#pragma unroll
		for (int jl = 1; jl <= 5; ++jl) {
			const int sz = 1 << jl, a = lane & ~(sz - 1);
			const int runs = (int)L.cH[a + sz] - (int)L.hB[a] + 1;
			if (runs <= RB3_RLE_CODES && level == jl - 1) level = jl;
		}
	}
	const bool sstart = lane < RB3_GRP_WINS && (lane & ((1 << level) - 1)) == 0;
	const uint32_t mask = (uint32_t)__ballot(sstart);
	const int nslots = __popc(mask);
	if (nslots > RB3_RG_MAXSLOTS) { RB3_REB_WHY(5); return false; }
	if (sstart) L.slotA[__popc(mask & ((1u << lane) - 1u))] = (uint32_t)lane;
	if (lane == 0) L.slotA[nslots] = RB3_GRP_WINS;
	wave_sync();
	RB3_REB_T(6);
#if defined(RB3_ABL) && RB3_ABL == 4
	return true;
#endif
	// ---- all slots of the group ----
	int TC = 0; // run codes of the group
	uint32_t bpm = 0; // slots that are single windows
	{
		uint32_t nc = 0;
		bool single = false;
		if (lane < nslots) {
			const uint32_t a = L.slotA[lane], an = L.slotA[lane + 1];
			single = an - a == 1u;
			if (!single) nc = (uint32_t)L.cH[an] - ((uint32_t)L.hB[a] - 1u);
		}
		const uint32_t inc = wave_incl_scan(nc);
		if (lane <= nslots && lane <= RB3_RG_MAXSLOTS) L.sbase[lane] = inc - nc;
		TC = (int)wave_read(inc, 63);
		bpm = (uint32_t)__ballot(single);
	}
	wave_sync();
	for (uint32_t m = bpm; m; m &= m - 1u) { // a single window: its symbols from the (few) runs that intersect it, as bit planes
		const int si = __ffs(m) - 1;
		const uint32_t a = L.slotA[si], x0 = a << RB3_WIN_BITS;
		const int t0 = (int)L.hB[a] - 1, nt = (int)L.cH[a + 1] - t0;
		uint32_t *pw = &L.stage[si * 24];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const uint32_t x = x0 + 64u * u + (uint32_t)lane;
			const uint32_t sy = H[t0 + lb_head(H + t0 + 1, nt - 1, x + 1u)] & 7u; // the last head at or before x
#pragma unroll
			for (int p = 0; p < 3; ++p) {
				const uint64_t bm = __ballot((sy >> p) & 1u);
				if (lane == u * 3 + p) pw[(u * 3 + p) * 2] = (uint32_t)bm, pw[(u * 3 + p) * 2 + 1] = (uint32_t)(bm >> 32);
			}
#pragma unroll
			for (int s6 = 0; s6 < 6; ++s6) {
				const uint32_t pc = (uint32_t)__popcll(__ballot(sy == (uint32_t)s6));
				if (lane == s6) atomicAdd(&L.scnt[si * 8 + s6], pc);
			}
		}
	}
	for (int c = 0; c * 64 < TC; ++c) { // one lane per run code
		const int x = c * 64 + lane;
		if (x < TC) {
			int si = 0;
#pragma unroll
			for (int q = 1; q < RB3_RG_MAXSLOTS; ++q) si += (q < nslots && (int)L.sbase[q] <= x) ? 1 : 0; // (bit-plane slots have no codes: equal bases, the last one wins)
			const uint32_t a = L.slotA[si], an = L.slotA[si + 1];
			const uint32_t x0 = a << RB3_WIN_BITS, x1 = an << RB3_WIN_BITS;
			const int ci = x - (int)L.sbase[si], t = (int)L.hB[a] - 1 + ci;
			const uint32_t h = H[t], hn = H[t + 1];
			const uint32_t st = (h >> 3) < x0 ? x0 : h >> 3, en = (hn >> 3) > x1 ? x1 : hn >> 3;
			((uint16_t*)L.stage)[si * RB3_RLE_CODES + ci] = (uint16_t)RB3_RUN_CODE(en - x0, h & 7u); // (cumulative: where the run ends in its slot)
			atomicAdd(&L.scnt[si * 8 + (h & 7u)], en - st);
		}
	}
	wave_sync();
	{ // headers and payload: lane = slot * 8 + slice
		const int si = lane >> 3, j = lane & 7;
		if (si < nslots) {
			const uint32_t a = L.slotA[si], an = L.slotA[si + 1];
			uint32_t hq;
			if (j == 0) hq = a << RB3_WIN_BITS | (an - a > 1u ? RB3_SLOT_RLE : 0u);
			else if (j == 7) hq = (an - a) << RB3_WIN_BITS;
			else {
				hq = 0u;
				for (int q = 0; q < si; ++q) hq += L.scnt[q * 8 + j - 1];
			}
			const uint32_t *pw = &L.stage[si * 24];
			uint4 v;
			v.x = hq;
			if (an - a > 1u) { // (the codes behind the last run repeat its end: the slot's symbols)
				const uint32_t tot = (an - a) << RB3_WIN_BITS;
				v.y = run_fill_unused(pw[j * 3], tot), v.z = run_fill_unused(pw[j * 3 + 1], tot), v.w = run_fill_unused(pw[j * 3 + 2], tot);
			} else v.y = pw[(j >> 1) * 6 + 0 + (j & 1)], v.z = pw[(j >> 1) * 6 + 2 + (j & 1)], v.w = pw[(j >> 1) * 6 + 4 + (j & 1)]; // same word order as wplane
			gslots[((int64_t)g * RB3_RG_MAXSLOTS + si) * 8 + j] = v;
		}
	}
	// ---- the group's record for the scan: symbol counts, slot count, slot-start mask (as k_decide) ----
	if (lane < 8) {
		uint32_t v = lane == 6 ? (uint32_t)nslots : mask;
		if (lane < 6) {
			v = 0u;
			for (int q = 0; q < nslots; ++q) v += L.scnt[q * 8 + lane];
		}
		gstat[g * 8 + lane] = v;
	}
	wave_sync();
	RB3_REB_T(7);
#ifdef RB3_PROF_REB
	tacc[8] += 1ull, tacc[9] += (unsigned long long)nb, tacc[10] += (unsigned long long)nR, tacc[11] += (unsigned long long)nslots;
#endif
	return true;
}

/* gkind[g]: 0 = rebuilt in run space (slots in the scratch, placed by k_place), 1 = not (the window kernels write it).  The groups a
 * tier cannot do are appended to lout (counter nlout), a wave's failures with one atomic: the list of the next tier (FROMLIST: a
 * small grid over lin[0 .. *nlin) instead of a look at every group), then of the LISTED window kernels. */
template<int RMAX, int NBMAX, bool FROMLIST>
__global__ void __launch_bounds__(64 * RB3_RG_WAVES, RMAX <= 384 ? 5 : 3) k_reb_group( // (waves per SIMD the LDS of the tier allows: registers must not be what limits them)
		IdxView old, const int64_t *pos, const uint8_t *b2, int64_t ntot, const int64_t *jw, int64_t ngrp,
		uint32_t *gstat, uint4 *gslots, uint8_t *gkind, const uint32_t *lin, const uint32_t *nlin, uint32_t *lout, uint32_t *nlout, const unsigned long long *skip)
{
	__shared__ RebLds<RMAX, NBMAX> lds_[RB3_RG_WAVES];
	if (RB3_REB_SKIP(skip)) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	RebLds<RMAX, NBMAX> &L = lds_[wave];
	const int64_t ustep = (int64_t)gridDim.x * RB3_RG_WAVES;
	const int64_t full = ntot >> RB3_GRP_BITS; // groups 0 .. full-1 are whole
	const int64_t nunits = FROMLIST ? (int64_t)*nlin : ngrp; // FROMLIST: the groups the tier before left (lin[0 .. *nlin)), else all groups
	int64_t u = (int64_t)blockIdx.x * RB3_RG_WAVES + wave;
	int64_t jn0 = 0, jn1 = 0; // the row range of the next group of this wave, requested one iteration ahead
	if (!FROMLIST && u < full) jn0 = jw[u * RB3_GRP_WINS], jn1 = jw[(u + 1) * RB3_GRP_WINS];
	int nfail = 0;
#ifdef RB3_PROF_REB
	unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
	for (; u < nunits; u += ustep) {
		int64_t g = u, j0 = jn0, j1 = jn1;
		if (FROMLIST) {
			g = (int64_t)lin[u];
			j0 = j1 = 0;
			if (g < full) j0 = jw[g * RB3_GRP_WINS], j1 = jw[(g + 1) * RB3_GRP_WINS];
		} else if (u + ustep < full) jn0 = jw[(u + ustep) * RB3_GRP_WINS], jn1 = jw[(u + ustep + 1) * RB3_GRP_WINS];
#ifdef RB3_PROF_REB
		const bool ok = reb_group_one<RMAX, NBMAX>(old, pos, b2, ntot, j0, j1, g, lane, L, gstat, gslots, tacc);
#else
		const bool ok = reb_group_one<RMAX, NBMAX>(old, pos, b2, ntot, j0, j1, g, lane, L, gstat, gslots);
#endif
		if (lane == 0) gkind[g] = ok ? 0 : 1;
		if (!ok) { // handed on: to the next tier's list, a wave's failures with one atomic
			if (lane == 0) L.fail[nfail] = (uint32_t)g;
			if (++nfail == RB3_RG_FAILBUF) {
				wave_sync();
				uint32_t o = 0;
				if (lane == 0) o = atomicAdd(nlout, (uint32_t)RB3_RG_FAILBUF);
				o = wave_read(o, 0);
				if (lane < RB3_RG_FAILBUF) lout[o + lane] = L.fail[lane];
				nfail = 0;
			}
		}
		wave_sync();
	}
	if (nfail > 0) {
		wave_sync();
		uint32_t o = 0;
		if (lane == 0) o = atomicAdd(nlout, (uint32_t)nfail);
		o = wave_read(o, 0);
		if (lane < nfail) lout[o + lane] = L.fail[lane];
	}
#ifdef RB3_PROF_REB
	if (lane == 0) for (int q = 0; q < 12; ++q) atomicAdd(&g_reb_prof[q], tacc[q]);
#endif
}

/* the groups rebuilt in run space: directory entry + slots from the scratch to their final places (the groups of the
 * window kernels are written by k_pass2w).  Eight lanes per group. */
__global__ void __launch_bounds__(256) k_place(const uint8_t *gkind, const uint32_t *gstat, const uint64_t *gpre, const uint64_t *tot, const uint4 *gslots,
		rb3_grp_t *grp, uint4 *slot16, int64_t ngrp, int64_t nwin, int64_t ntot, const unsigned long long *skip,
		const uint32_t *nglist, uint32_t lcap, uint64_t slot_cap, int64_t abs_lim = RB3_ABS_LIMIT)
{
	if (RB3_REB_SKIP(skip)) return;
	if (*nglist > lcap || tot[6] > slot_cap) return; // see k_decide / k_pass2w: the host does the rebuild again
#ifdef RB3_ABL
	return; // (kernel ablation builds leave the group records undefined)
#endif
	const int j = threadIdx.x & 7;
	const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
	if (g >= ngrp || gkind[g] != 0) return;
	const uint32_t gs = gstat[g * 8 + j];     // lane 6: slots, lane 7: mask
	const uint64_t gp = gpre[g * 8 + j];      // lanes 0..5: symbols before the group, lane 6: slots before the group
	uint64_t cb = 0;                          // C[j] of the merged BWT (lanes 0..5)
	for (int a = 0; a < j && a < 6; ++a) cb += tot[a];
	const uint32_t ns = (uint32_t)__shfl((int)gs, 6, 8), mask = (uint32_t)__shfl((int)gs, 7, 8);
	const uint64_t slot0 = (uint64_t)(uint32_t)__shfl((int)(uint32_t)gp, 6, 8) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(gp >> 32), 6, 8) << 32;
	((uint64_t*)grp)[g * 8 + j] = j < 6 ? cb + gp : j == 6 ? (uint64_t)(uint32_t)slot0 | (uint64_t)mask << 32 : 0ull;
	// headers that carry the whole LF base (RB3_ABS_HEADERS): header lane j holds symbol j-1, whose base sits in lane j-1
	const uint32_t below = (uint32_t)__shfl((int)(uint32_t)(cb + gp), (j + 7) & 7, 8);
	const uint32_t add = (RB3_ABS_HEADERS(ntot, abs_lim) && j >= 1 && j <= 6) ? below : 0u;
	for (uint32_t si = 0; si < ns && si < RB3_RG_MAXSLOTS; ++si) {
		uint4 v = gslots[(g * RB3_RG_MAXSLOTS + si) * 8 + j];
		v.x += add;
		slot16[(slot0 + si) * 8 + j] = v;
	}
}

/* ----------------------------------------------------------------------------------------- */
/* text-regular walkers from the BWT alone (the reference's signature: no suffix array)         */
/* ----------------------------------------------------------------------------------------- */

/* rb3_fmi_merge_plain(r, len, seq) gets the partial BWT and nothing else, and a batch of two long strings is two chains.
 * Walkers in the middle of the strings need rows at (roughly) regular TEXT distances, i.e. samples of the inverse suffix
 * array, which the BWT only yields through its own LF walk.  That walk is done once, sparsely (sampled list ranking,
 * Helman-JaJa): SPLITTERS -- the sentinel rows and every 2^S-th row, random text positions at mean distance 2^S -- walk
 * the batch's LF words to the next splitter (k_b2_walk; ~2^S ln(#splitters) dependent 8-byte loads on the critical path,
 * no rank on B1), pointer jumping gives every splitter its distance D from the start of its string (k_ssa_jump), and
 * from each window of RB3_B2_W text positions the splitter closest to the window's start becomes a walker if it lies in
 * the first two thirds (k_b2_pick / k_b2_list): gaps between walkers are then RB3_B2_W +- a few 2^S and never below 128, which is
 * what k_chain's plain-store records need (see "Concurrency").  Everything stays on the device; the list is read by
 * k_chain through its device-side length. */
#define RB3_SSA_END  (1ull << 63)  /* link word: the sublist ends at the start of its string (shared with the sampled suffix array) */
#define RB3_B2_W     384           /* text distance between walkers (a multiple of it where the stretch table would not take this many walkers' events) */
                                   /* a splitter qualifies if it lies in the first W - RB3_B2_MINSEG positions of its window (W = 384: 256) */
#define RB3_B2_MINSEG 128          /* shortest segment of any walker (what plain-store records need, see k_chain) */
#define RB3_B2_EMPTY (~0ull)

__global__ void k_ssa_jump(int64_t nsp, const uint64_t *in, uint64_t *out); // (pointer jumping over splitters: defined with the sampled suffix array)

/* pointer jumping, three hops per round (links of the OLD table only, so one round multiplies the reach of every link by
 * four): half the launches of the doubling form, and a launch costs more here than two more dependent gathers (seven hops per
 * round, a third fewer launches, was measured too: 0.05 ms slower on the bench step).  (All rounds
 * in one launch with grid barriers -- 512 resident blocks, release/acquire fences at agent scope -- was measured: 0.6 ms
 * instead of 0.2 ms for the ten rounds; the per-round L2 write-back and invalidate cost more than the launches.) */
__global__ void __launch_bounds__(256) k_b2_jump4(int64_t nsp, const uint64_t *in, uint64_t *out)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nsp) return;
	ulonglong2 r = ((const ulonglong2*)in)[p];
#pragma unroll
	for (int hop = 0; hop < 3; ++hop) {
		if ((r.x & RB3_SSA_END) || r.x >= (uint64_t)nsp) break;
		const ulonglong2 q = ((const ulonglong2*)in)[r.x];
		r.x = q.x, r.y += q.y;
	}
	((ulonglong2*)out)[p] = r;
}

/* mode[0]: 0 = splitters, 1 = strings are short (one walker per string), 2 = more strings than the list can take */
__global__ void __launch_bounds__(256) k_b2_mode(const uint64_t *tot2, int64_t n2, int64_t m2cap, unsigned long long *mode, int64_t W)
{
	if (threadIdx.x || blockIdx.x) return;
	const int64_t m2 = (int64_t)tot2[0];
	mode[0] = m2 <= 0 || m2 > m2cap ? 2ull : (n2 / m2 <= 4 * W ? 1ull : 0ull);
}

__global__ void __launch_bounds__(256) k_b2_walk(const int64_t *roww, int64_t n2, const uint64_t *tot2, int S, const unsigned long long *mode, uint64_t *lnk)
{
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0], msk = (1LL << S) - 1;
	const int64_t nsp = m2 + ((n2 - m2 + msk) >> S);
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nsp) return;
	int64_t r = p < m2 ? p : m2 + ((p - m2) << S);
	uint64_t steps = 0, nxt;
	for (;;) {
		const uint64_t w = (uint64_t)roww[r];
		if ((w & 7u) == 0u) { nxt = RB3_SSA_END | (uint64_t)RB3_ROW_NEXT(w); break; } // the first suffix of a string: its $ has a dense number
		r = RB3_ROW_NEXT(w), ++steps;
		if (r >= m2 && ((r - m2) & msk) == 0) { nxt = (uint64_t)(m2 + ((r - m2) >> S)); break; }
	}
	lnk[2 * p] = nxt, lnk[2 * p + 1] = steps;
}

/* slen[s] = rows of string s (the LF steps of its sentinel row's splitter to the start of the string, + 1) */
__global__ void __launch_bounds__(256) k_b2_strings(const uint64_t *tot2, const unsigned long long *mode, const uint64_t *lnk, uint64_t *slen)
{
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0];
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m2; j += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t e = lnk[2 * j];
		if (e & RB3_SSA_END) slen[e & ~RB3_SSA_END] = lnk[2 * j + 1] + 1;
	}
}

/* slen[] -> its exclusive prefix in place, total behind the last entry (one workgroup; the strings are long, so there are few) */
__global__ void __launch_bounds__(1024) k_b2_scan(const uint64_t *tot2, const unsigned long long *mode, uint64_t *slen)
{
	__shared__ uint64_t part[1024];
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0];
	const int t = threadIdx.x;
	const int64_t per = (m2 + 1023) / 1024, a = t * per, b = a + per < m2 ? a + per : m2;
	uint64_t sum = 0;
	for (int64_t i = a; i < b; ++i) sum += slen[i];
	part[t] = sum;
	__syncthreads();
	if (t == 0) { uint64_t run = 0; for (int i = 0; i < 1024; ++i) { const uint64_t v = part[i]; part[i] = run; run += v; } slen[m2] = run; }
	__syncthreads();
	uint64_t run = part[t];
	for (int64_t i = a; i < b; ++i) { const uint64_t v = slen[i]; slen[i] = run; run += v; }
}

/* Round 6: the TEXT-ORDER WORDS of a batch from its BWT alone (the reference's signature, rb3_fmi_merge_plain(len, bwt), fm-index.c:279): what a suffix sorter
 * hands over for free -- tw[t] = row of the suffix at text position t << 3 | the symbol before it -- is the inverse of the BWT, and the sparse LF walk above
 * has already done the sequential part: after the jumping rounds every splitter knows its string and its distance from the string's start.  So every
 * splitter walks its stretch of rows ONCE MORE and writes their words at its text positions; the batch is then merged exactly like one that came with
 * its inverse suffix array (k_chain<..., TEXT>: streamed words, the common step), instead of by walkers that chase row words.
 * k_b2_strings2: string j (its sentinel is row j) has slen2[j] rows; its first suffix links to the dense number X of the string before it: sidx[X] = j. */
__global__ void __launch_bounds__(256) k_b2_strings2(const uint64_t *tot2, const unsigned long long *mode, const uint64_t *lnk, uint64_t *slen2, uint64_t *sidx)
{
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0];
	for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m2; j += (int64_t)gridDim.x * blockDim.x) {
		const uint64_t e = lnk[2 * j];
		slen2[j] = lnk[2 * j + 1] + 1;
		if ((e & RB3_SSA_END) && (e & ~RB3_SSA_END) < (uint64_t)m2) sidx[e & ~RB3_SSA_END] = (uint64_t)j;
	}
}

/* gbase2 = the exclusive prefix of slen2 (k_b2_scan): string j lies at text positions [gbase2[j], gbase2[j + 1]), its sentinel last -- the batch's own order */
__global__ void __launch_bounds__(256) k_b2_tw(const int64_t *roww, int64_t n2, const uint64_t *tot2, int S, const unsigned long long *mode, const uint64_t *lnk, const uint64_t *gbase2,
		const uint64_t *sidx, uint64_t *tw, unsigned long long *bad)
{
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0], msk = (1LL << S) - 1;
	const int64_t nsp = m2 + ((n2 - m2 + msk) >> S);
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nsp) return;
	const uint64_t e = lnk[2 * p], D = lnk[2 * p + 1];
	if (!(e & RB3_SSA_END) || (e & ~RB3_SSA_END) >= (uint64_t)m2) { atomicAdd(bad, 1ull); return; } // (cannot be after the jumping rounds)
	const uint64_t j = sidx[e & ~RB3_SSA_END];
	if (j >= (uint64_t)m2 || gbase2[j] + D >= (uint64_t)n2) { atomicAdd(bad, 1ull); return; }
	int64_t r = p < m2 ? p : m2 + ((p - m2) << S), G = (int64_t)(gbase2[j] + D);
	for (;;) { // the same stretch as k_b2_walk's: down to the row in front of the next splitter, or to the first suffix of the string
		const uint64_t w = (uint64_t)roww[r];
		tw[G] = (uint64_t)r << 3 | (w & 7u); // (written as a stream -- nt -- the one-genome merge took 1.71 instead of 1.56 ms)
		if ((w & 7u) == 0u) break;
		r = RB3_ROW_NEXT(w), --G;
		if ((r >= m2 && ((r - m2) & msk) == 0) || G < 0) break;
	}
}

/* per window of RB3_B2_W text positions (concatenated strings): the qualifying splitter closest to the window's start */
__global__ void __launch_bounds__(256) k_b2_pick(int64_t n2, const uint64_t *tot2, int S, const unsigned long long *mode, const uint64_t *lnk, const uint64_t *gbase, unsigned long long *bucket, int64_t W)
{
	if (mode[0] != 0) return;
	const int64_t m2 = (int64_t)tot2[0], msk = (1LL << S) - 1;
	const int64_t nsp = m2 + ((n2 - m2 + msk) >> S);
	const int64_t p = m2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nsp) return;
	const uint64_t e = lnk[2 * p], D = lnk[2 * p + 1];
	if (!(e & RB3_SSA_END)) return; // (cannot be after the jumping rounds)
	const uint64_t sid = e & ~RB3_SSA_END;
	if (sid >= (uint64_t)m2) return;
	const uint64_t g0 = gbase[sid], len = gbase[sid + 1] - g0, G = g0 + D;
	if (D == 0 || D + 1 + RB3_B2_MINSEG > len) return; // the sentinel walker's own segment stays >= RB3_B2_MINSEG steps
	const uint64_t off = G % (uint64_t)W;
	if (off < (uint64_t)(W - RB3_B2_MINSEG)) atomicMin(&bucket[G / (uint64_t)W], (unsigned long long)(off << 48 | (uint64_t)p)); // (gaps stay >= RB3_B2_MINSEG)
}

/* A pick lies `off` text positions behind the start of its window (geometric, mean 2^S, the odd one beyond 100), and a
 * wave of k_chain lasts as long as its longest walker while walkers of different lengths fall out of step (exact spacing 384:
 * 0.46 ms, this jitter: 0.71 ms on the bench workload).  So the pick WALKS its `off` LF steps down to the window start (the
 * batch's own LF words: dependent 8-byte loads, no rank) and the walker starts exactly there -- unless that would leave the
 * pick's string (then it stays where it is).  b2_refined_D: the distance from the start of its string at which a pick ends up. */
__device__ __forceinline__ uint64_t b2_refined_D(uint64_t D, uint64_t g0, int64_t W)
{
	const uint64_t off = (g0 + D) % (uint64_t)W;
	return D > off ? D - off : D;
}

/* the walker list: slot b < nbk = the pick of window b (row -1: none), slot nbk + j = the sentinel row j; nsteps = text distance
 * to the next walker on the left in the same string.  nwalk[0] = slots in use. */
__global__ void __launch_bounds__(256) k_b2_list(int64_t n2, const uint64_t *tot2, int S, const unsigned long long *mode, const uint64_t *lnk, const uint64_t *gbase,
		const unsigned long long *bucket, int64_t nbk, Walker *wl, unsigned long long *nwalk, const int64_t *roww, int64_t W)
{
	const int64_t m2 = (int64_t)tot2[0];
	const unsigned long long md = mode[0];
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
	if (md == 2) { if (t == 0) nwalk[0] = 0; return; }
	if (md == 1) { // short strings: one walker per string
		if (t == 0) nwalk[0] = (unsigned long long)m2;
		for (int64_t j = t; j < m2; j += nt) { Walker w; w.row = j, w.ka0 = -2, w.nsteps = INT64_MAX / 2, w.flags = 0; wl[j] = w; }
		return;
	}
	if (t == 0) nwalk[0] = (unsigned long long)(nbk + m2);
	for (int64_t u = t; u < nbk + m2; u += nt) {
		Walker w;
		w.row = -1, w.ka0 = -1, w.nsteps = INT64_MAX / 2, w.flags = 0;
		uint64_t sid, D, G;
		int64_t b;
		if (u < nbk) {
			const unsigned long long v = bucket[u];
			if (v == RB3_B2_EMPTY) { wl[u] = w; continue; }
			const int64_t p = (int64_t)(v & 0xFFFFFFFFFFFFull);
			w.row = m2 + ((p - m2) << S);
			sid = lnk[2 * p] & ~RB3_SSA_END, D = lnk[2 * p + 1], b = u - 1;
			const uint64_t D2 = b2_refined_D(D, gbase[sid], W);
			for (uint64_t i = D2; i < D; ++i) w.row = RB3_ROW_NEXT((uint64_t)roww[w.row]); // (no sentinel on the way: D2 >= 1)
			D = D2, G = gbase[sid] + D;
		} else {
			const int64_t j = u - nbk;
			w.row = j, w.ka0 = -2;
			sid = lnk[2 * j] & ~RB3_SSA_END, D = lnk[2 * j + 1], G = gbase[sid] + D, b = (int64_t)(G / (uint64_t)W);
		}
		// the next walker on the left: the pick of the nearest window below that lies in the same string
		for (; b >= 0 && (uint64_t)(b + 1) * (uint64_t)W > gbase[sid]; --b) {
			const unsigned long long v = bucket[b];
			if (v == RB3_B2_EMPTY) continue;
			const int64_t q = (int64_t)(v & 0xFFFFFFFFFFFFull);
			if ((lnk[2 * q] & ~RB3_SSA_END) != sid) continue; // a window shared with the neighbouring string
			const uint64_t Dq = b2_refined_D(lnk[2 * q + 1], gbase[sid], W);
			if (Dq < D) { w.nsteps = (int64_t)(D - Dq); break; }
		}
		wl[u] = w;
	}
}

/* ----------------------------------------------------------------------------------------- */
/* interval-sharded index: one LF step of many chains per launch (north_star)  */
/* ----------------------------------------------------------------------------------------- */

/* The accumulated BWT is cut into contiguous INTERVALS of positions, one per GPU; a handle holds the block array of its
 * interval only.  rank over the whole BWT = rank inside the interval + the symbol totals of the intervals before it -- the
 * analogue of the walk over the six ropes' totals in mr_rank2a (mrope.c:76-88) -- so with adj[c] = C[c] + (symbols c in the
 * intervals before) - C_local[c] the LF step of rb3_mg_rank1_plain (fm-index.c:171-173) is LF_local(c, ka - start) + adj[c].
 * A chain STATE (text position tp of the current suffix, its insertion point ka) lives on the GPU whose interval contains
 * ka.  k_sh_step advances every state of this GPU by one symbol: record ka for the suffix's row, next ka, and the interval
 * that owns it; k_sh_scatter groups the new states by destination for the all-to-all.  Chains advance in lock step, one
 * collective per symbol: for batches of short strings (the kt_for over millions of reads, fm-index.c:217-224). */
#define RB3_SH_MAXIV 64

struct ShState { int64_t tp, ka; };
struct ShArgs {
	int64_t adj[6];
	int64_t bounds[RB3_SH_MAXIV + 1]; // interval i = [bounds[i], bounds[i+1])
	int64_t iv_start;
	int n_iv;
};

/* counts[d], d = 0..n_iv-1: states that go to interval d; counts[n_iv]: chains that ended (the start of a string was reached) */
__global__ void __launch_bounds__(256) k_sh_step(IdxView ix, ShArgs a, int64_t n, const ShState *in, const uint64_t *tw, int64_t *ka_rec,
		ShState *out, int32_t *dest, unsigned long long *counts, unsigned long long *bad)
{
	__shared__ uint32_t lc[RB3_SH_MAXIV + 1];
	const int lane = threadIdx.x & 63, j = lane & 7;
	for (int i = threadIdx.x; i <= a.n_iv; i += blockDim.x) lc[i] = 0u;
	__syncthreads();
	const int64_t nq8 = (int64_t)gridDim.x * (blockDim.x >> 3);
	for (int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; q < n; q += nq8) {
		const ShState st = in[q];
		const uint64_t x = tw[st.tp];
		const int64_t kb = (int64_t)(x >> 3);
		const int c = (int)(x & 7u);
		int64_t k = st.ka - a.iv_start;
		if (k < 0 || k > ix.n) { if (j == 0) atomicAdd(bad, 1ull); k = k < 0 ? 0 : ix.n; } // (a state routed to the wrong interval: cannot be)
		if (j == 0) ka_rec[kb] = st.ka;
		int d = a.n_iv;
		ShState nx;
		nx.tp = st.tp - 1, nx.ka = 0;
		if (c != 0) { // (wave-uniform control flow is not required: the rank helpers only talk inside an octet)
			RankLoad r;
			oct_rank_issue(ix, k, j, r);
			nx.ka = oct_rank_finish(r, c, j, ix.abs) + a.adj[c];
			d = 0;
			for (int i = 1; i < a.n_iv; ++i) d += a.bounds[i] <= nx.ka ? 1 : 0;
		}
		if (j == 0) {
			out[q] = nx, dest[q] = d;
			atomicAdd(&lc[d], 1u);
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i <= a.n_iv; i += blockDim.x)
		if (lc[i]) atomicAdd(&counts[i], (unsigned long long)lc[i]);
}

/* send[off[d] ...] = the states with dest d (any order inside a destination); cursor[] zeroed by the caller */
struct ShOffsets { int64_t off[RB3_SH_MAXIV + 1]; int n_iv; };
__global__ void __launch_bounds__(256) k_sh_scatter(ShOffsets o, int64_t n, const ShState *out, const int32_t *dest, ShState *send, unsigned long long *cursor)
{
	__shared__ uint32_t lc[RB3_SH_MAXIV + 1];
	__shared__ unsigned long long lb[RB3_SH_MAXIV + 1];
	for (int i = threadIdx.x; i <= o.n_iv; i += blockDim.x) lc[i] = 0u;
	__syncthreads();
	const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int d = -1;
	uint32_t mine = 0;
	if (q < n) {
		d = dest[q];
		if (d < o.n_iv) mine = atomicAdd(&lc[d], 1u); else d = -1; // ended chains are not sent anywhere
	}
	__syncthreads();
	for (int i = threadIdx.x; i < o.n_iv; i += blockDim.x)
		lb[i] = lc[i] ? atomicAdd(&cursor[i], (unsigned long long)lc[i]) : 0ull;
	__syncthreads();
	if (d >= 0) send[o.off[d] + (int64_t)lb[d] + mine] = out[q];
}

/* merged position inside the interval of the rows [jlo, jlo + n) that landed in it: pos[r] = (ka - start) + r */
__global__ void __launch_bounds__(256) k_sh_localpos(int64_t jlo, int64_t n, const int64_t *ka_rec, int64_t iv_start, int64_t *pos, unsigned long long *bad)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n) return;
	const int64_t ka = ka_rec[jlo + r];
	if (ka < iv_start) { atomicAdd(bad, 1ull); pos[r] = RB3_UNSET; return; } // unset (-1) or routed wrongly
	pos[r] = ka - iv_start + r;
}

/* The same walk driven from inside the library (rb3gpu_sh_merge): ONE kernel per lock-step round, no scatter pass and no array
 * over all rows of the batch.  State q of a round that starts with `done` rows recorded on this GPU appends the (row, insertion
 * point) pair of its suffix at rec[done + q] -- arrival order, no counter; the rows that land in an interval are a contiguous
 * range of the batch's rows because pos[] is increasing, k_sh_place sorts them out at the end -- and writes its next state
 * straight into the send region of the interval that owns it: region d = send + d * stride, a slot from the cursor cnt[d]
 * (one atomic per block, destination and 32 states).  After the launch the cursors ARE the split sizes of the all-to-all;
 * cnt[n_iv] counts the chains that reached the start of their string.  The counters of the other parity are cleared for the
 * next round on the way (the host has read them before this launch), so a round is one launch and one 8-byte-per-rank read-back. */
struct ShRec { int64_t kb, ka; };
/* S states per octet, their loads issued stage by stage (states, text-order words, directory words, slots: S independent requests in
 * flight per stage instead of a chain of four per state), and ONE cursor atomic per block and destination for its 32 S states: the
 * cursors are single words that every block adds to and needs the answer from, and such an atomic takes ~12 ns of its L2 channel --
 * with one per 32 states the kernel ran at 2.5 G steps/s whatever the number of chains (measured: 398 us per 10^6 chains). */
/* tprev != NULL (the batch sharded, VERDICT r4 "a sharded batch"): the device holds the symbol before every text position, one byte each,
 * instead of the text-order words (8 bytes each) -- what a step needs of the batch is that symbol only; the record is then
 * (text position, insertion point) and the row is looked up at the end, by the rank that holds that part of the inverse suffix array. */
/* BS threads per block: every block makes ONE returning atomic per destination on the round's cursor, and such atomics on one word take ~12 ns
 * each one after the other -- 7800 blocks of 256 threads at 2 M chains are 94 us of them per round; blocks of 1024 threads are a quarter of that */
/* PEER ROUNDS (round 6; ranks that address each other's device memory: threads of one process with peer access over xGMI, rb3gpu_group_*).  The
 * next state of a chain is written straight into the RECEIVE buffer of the rank that owns it: rank d's buffer has one region per source rank (stride
 * states each), a source takes the places in ITS region from its own cursors (the same local atomics as with send regions: nothing returns over a
 * link) and stores the state there -- 16 bytes, fire and forget.  The block that finishes last stores the source's totals into the destinations'
 * tables of incoming counts (cin[d][source], one 8-byte word per pair and round), and the next round's kernel on d finds its states as world
 * regions of cin_mine[s] states each.  A round is ONE kernel per rank and nothing goes through the host: no read-back of split sizes, no all-gather,
 * no all-to-all (rb3gpu.hip, sh_merge_impl); between two rounds the streams wait for each other's events (rb3gpu_comm_t.stream_barrier) -- what one
 * kernel stored is there for the kernels queued behind such a wait, on whichever device.  rec_cap: records this rank has room for (nobody knows
 * beforehand how many rows land in an interval); a round that would overrun it counts into bad[1] and the host does the merge again its old way. */
#define RB3_SH_MAXPEER 8
struct ShPeers {
	ShState *dst[RB3_SH_MAXPEER];            // receive buffer of rank d for the round behind this one (on device d)
	unsigned long long *cin[RB3_SH_MAXPEER]; // rank d's incoming counts of that round: cin[d][source]
	const unsigned long long *cin_mine;      // this rank's incoming counts of THIS round, by source
	unsigned long long *done;                // blocks of this launch that have taken their places (cleared by the last one)
	int64_t stride, rec_cap;
	int on, rank;
};
template<int S, int BS = 256>
__global__ void __launch_bounds__(BS) k_sh_round(IdxView ix, ShArgs a, int64_t n, const ShState *in, const uint64_t *tw, ShRec *rec,
		ShState *send, int64_t stride, unsigned long long *cnt, unsigned long long *cnt_next, unsigned long long *bad, const uint8_t *tprev = nullptr,
		const unsigned long long *n_dev = nullptr, unsigned long long *rowbase = nullptr, ShPeers peers = ShPeers())
{
	// n_dev, rowbase (ONE interval: the rounds run back to back, nothing goes to the host between them): the number of states of this round is what
	// the round before counted for interval 0 (*n_dev; `n` is then only an upper bound that sized the grid), its records go behind those of the
	// rounds before (rowbase[0]), and block 0 leaves the base of the next round in rowbase[1]
	__shared__ uint32_t lc[RB3_SH_MAXIV + 1];
	__shared__ unsigned long long lb[RB3_SH_MAXIV + 1];
	const int j = threadIdx.x & 7;
	__shared__ unsigned long long pre[RB3_SH_MAXPEER + 1]; // peer rounds: the states of source s are [pre[s], pre[s + 1]) of this round
	if (n_dev != nullptr || peers.on) { // (ONE thread of the block fetches the two words: a million threads asking for the same line kept its L2 channel busy for longer than the round's work)
		__shared__ unsigned long long nb[2];
		if (threadIdx.x == 0) {
			if (peers.on) {
				unsigned long long t = 0;
				for (int q = 0; q < a.n_iv; ++q) pre[q] = t, t += peers.cin_mine[q];
				for (int q = a.n_iv; q <= RB3_SH_MAXPEER; ++q) pre[q] = t;
				nb[0] = t;
			} else nb[0] = *n_dev;
			nb[1] = rowbase[0];
		}
		__syncthreads();
		const int64_t nd = (int64_t)nb[0];
		n = nd < n ? nd : n;
		if (peers.on && (int64_t)nb[1] + n > peers.rec_cap) { // more rows land here than this rank made room for: nothing is written, the host does the merge again
			if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(bad + 1, 1ull);
			n = 0;
		}
		rec += nb[1];
		if (blockIdx.x == 0 && threadIdx.x == 0) rowbase[1] = nb[1] + (unsigned long long)n;
		if ((int64_t)blockIdx.x * (BS / 8) * S >= n && blockIdx.x != 0) return; // (block 0 stays: it clears the counters of the next round)
	}
	// (peer rounds: cursor d lies at word 16 d of its set -- a line of its own, so that the returning atomics of different destinations do not queue
	// in one L2 channel, ~12 ns each -- and the count of ended chains at word 8)
	if (blockIdx.x == 0 && threadIdx.x <= (peers.on ? 16 * RB3_SH_MAXPEER - 1 : RB3_SH_MAXIV)) cnt_next[threadIdx.x] = 0ull;
	for (int i = threadIdx.x; i <= a.n_iv; i += blockDim.x) lc[i] = 0u;
	__syncthreads();
	const int64_t q0 = ((int64_t)blockIdx.x * (BS / 8) + (threadIdx.x >> 3)) * S;
	ShState st[S];
	uint64_t x[S];
	RankLoad r[S];
	int d[S];
	uint32_t mine[S];
	int src = 0; // peer rounds: the source whose region holds state q0 (the S states of an octet follow each other: the search is done once)
	if (peers.on) {
#pragma unroll
		for (int e = 1; e < RB3_SH_MAXPEER; ++e) src += pre[e] <= (unsigned long long)q0 && e < a.n_iv ? 1 : 0;
	}
#pragma unroll
	for (int s = 0; s < S; ++s) {
		st[s].tp = 0, st[s].ka = a.iv_start;
		if (q0 + s < n && peers.on) {
			const unsigned long long q = (unsigned long long)(q0 + s);
			while (src + 1 < a.n_iv && pre[src + 1] <= q) ++src;
			st[s] = in[(int64_t)src * peers.stride + (int64_t)(q - pre[src])];
		} else if (q0 + s < n) st[s] = in[q0 + s];
	}
#pragma unroll
	for (int s = 0; s < S; ++s) x[s] = q0 + s >= n ? 0ull : tprev ? ((uint64_t)st[s].tp << 3 | (uint64_t)tprev[st[s].tp]) : tw[st[s].tp];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		int64_t k = st[s].ka - a.iv_start;
		if (k < 0 || k > ix.n) { if (j == 0) atomicAdd(bad, 1ull); k = k < 0 ? 0 : ix.n; } // (a state routed to the wrong interval: cannot be)
		if (j == 0 && q0 + s < n) { ShRec t; t.kb = (int64_t)(x[s] >> 3), t.ka = st[s].ka; rec[q0 + s] = t; }
		oct_rank_issue_grp(ix, k, j, r[s]); // (also for a chain that ends here or a state past the end: the address is valid, the result unused)
	}
#pragma unroll
	for (int s = 0; s < S; ++s) oct_rank_issue_slot(ix, j, r[s]);
#pragma unroll
	for (int s = 0; s < S; ++s) {
		const int c = (int)(x[s] & 7u);
		d[s] = -1, mine[s] = 0;
		st[s].tp -= 1;
		if (q0 + s < n) {
			d[s] = a.n_iv;
			if (c != 0) {
				int64_t adj = a.adj[0];
#pragma unroll
				for (int e = 1; e < 6; ++e) adj = c == e ? a.adj[e] : adj; // (selects on scalars: an index that varies per lane would move the array to scratch)
				st[s].ka = oct_rank_finish(r[s], c, j, ix.abs) + adj;
				d[s] = 0;
				for (int i = 1; i < a.n_iv; ++i) d[s] += a.bounds[i] <= st[s].ka ? 1 : 0;
			}
			if (j == 0) mine[s] = atomicAdd(&lc[d[s]], 1u);
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i <= a.n_iv; i += blockDim.x)
		lb[i] = lc[i] ? atomicAdd(&cnt[peers.on ? (i < a.n_iv ? 16 * i : 8) : i], (unsigned long long)lc[i]) : 0ull;
	__syncthreads();
#pragma unroll
	for (int s = 0; s < S; ++s)
		if (j == 0 && d[s] >= 0 && d[s] < a.n_iv) {
			if (peers.on) { // into this rank's region of the owner's receive buffer
				ShState *p = peers.dst[0];
#pragma unroll
				for (int e = 1; e < RB3_SH_MAXPEER; ++e) p = d[s] == e ? peers.dst[e] : p; // (selects: an index that varies per lane would move the table to scratch)
				p[(int64_t)peers.rank * peers.stride + (int64_t)lb[d[s]] + mine[s]] = st[s];
			} else send[(int64_t)d[s] * stride + (int64_t)lb[d[s]] + mine[s]] = st[s];
		}
	if (peers.on) { // the block that takes its places last knows this rank's totals: they go to the owners' tables of incoming counts
		__shared__ int last;
		// (no fence: the cursor atomics above RETURNED their values before the barrier in front of the stores, so they are done where the last block's
		// atomic loads look for them; the states themselves are only read by kernels behind this one.  A __threadfence() here -- a release at agent
		// scope, i.e. a write-back of the XCD's L2 per block -- made a round of 2 M chains take 1150 us instead of 220.)
		__syncthreads();
		if (threadIdx.x == 0) {
			const int64_t per = (int64_t)(BS / 8) * S;
			const unsigned long long nact = n > 0 ? (unsigned long long)((n + per - 1) / per) : 1ull; // (the blocks that did not return at the top)
			last = atomicAdd(peers.done, 1ull) + 1ull == nact ? 1 : 0;
		}
		__syncthreads();
		if (last && threadIdx.x < a.n_iv) {
			const unsigned long long v = __hip_atomic_load(&cnt[16 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			unsigned long long *c = peers.cin[0];
#pragma unroll
			for (int e = 1; e < RB3_SH_MAXPEER; ++e) c = (int)threadIdx.x == e ? peers.cin[e] : c;
			__hip_atomic_store(&c[peers.rank], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		if (last && threadIdx.x == 0) *peers.done = 0ull;
	}
}

/* the pairs an interval collected -> merged positions inside the interval: the rows are [jlo, jlo + n) of the batch, row r lands
 * at (ka - start) + (r - jlo).  pos[] comes filled with RB3_UNSET; a row outside the range, recorded twice (then another one stays
 * unset) or routed wrongly shows in bad[0] here or in k_pos_check behind this kernel. */
__global__ void __launch_bounds__(256) k_sh_place(int64_t n, const ShRec *rec, int64_t jlo, int64_t iv_start, int64_t *pos, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const ShRec r = rec[i];
	const int64_t kb = r.kb - jlo;
	if (kb < 0 || kb >= n || r.ka < iv_start) { atomicAdd(bad, 1ull); return; }
	pos[kb] = r.ka - iv_start + kb;
}

/* ---- the batch sharded by text range: the rows of the records are looked up where that part of the inverse suffix array lives ---- */

/* the symbol before every text position, one byte each, from the text-order words (tw[t] = row << 3 | symbol before) */
__global__ void __launch_bounds__(256) k_tprev_from_tw(const uint64_t *tw, int64_t n, uint8_t *out)
{
	const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
	if (i >= n) return;
	if (i + 8 <= n) {
		uint64_t v = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) v |= (tw[i + k] & 7ull) << (8 * k);
		*(uint64_t*)(out + i) = v;
	} else
		for (int64_t k = i; k < n; ++k) out[k] = (uint8_t)(tw[k] & 7ull);
}

struct ShTextBounds { int64_t b[RB3_SH_MAXIV + 1]; int n; };

/* records (text position, insertion point) per owner of the text position: cnt[q] += ... */
__global__ void __launch_bounds__(256) k_sh_owner_count(int64_t n, const ShRec *rec, ShTextBounds tb, unsigned long long *cnt, unsigned long long *bad)
{
	__shared__ unsigned int lc[RB3_SH_MAXIV];
	if (threadIdx.x < RB3_SH_MAXIV) lc[threadIdx.x] = 0u;
	__syncthreads();
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const int64_t tp = rec[i].kb;
		if (tp < 0 || tp >= tb.b[tb.n]) atomicAdd(bad, 1ull);
		else {
			int d = 0;
			for (int q = 1; q < tb.n; ++q) d += tb.b[q] <= tp ? 1 : 0;
			atomicAdd(&lc[d], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x < tb.n && lc[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)lc[threadIdx.x]);
}

/* ... and into the send region of that owner (region q at q * stride; the order inside a region does not matter: the reply carries all a row needs) */
__global__ void __launch_bounds__(256) k_sh_owner_scatter(int64_t n, const ShRec *rec, ShTextBounds tb, int64_t stride, unsigned long long *cursor, ShRec *send)
{
	__shared__ unsigned int lc[RB3_SH_MAXIV];
	__shared__ unsigned long long lb[RB3_SH_MAXIV];
	if (threadIdx.x < RB3_SH_MAXIV) lc[threadIdx.x] = 0u;
	__syncthreads();
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	int d = -1;
	unsigned int mine = 0;
	ShRec r;
	r.kb = r.ka = 0;
	if (i < n) {
		r = rec[i];
		if (r.kb >= 0 && r.kb < tb.b[tb.n]) {
			d = 0;
			for (int q = 1; q < tb.n; ++q) d += tb.b[q] <= r.kb ? 1 : 0;
			mine = atomicAdd(&lc[d], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x < tb.n) lb[threadIdx.x] = lc[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)lc[threadIdx.x]) : 0ull;
	__syncthreads();
	if (d >= 0) send[(int64_t)d * stride + (int64_t)lb[d] + mine] = r;
}

/* the owner's answer: request i of source s (the requests lie packed by source, off[s] .. off[s + 1]) -> tw[tp] = row << 3 | symbol of the row,
 * into the send region of s (s * stride); the insertion point travels back untouched */
__global__ void __launch_bounds__(256) k_sh_lookup(int64_t n, const ShRec *req, ShTextBounds off, int64_t stride, const uint64_t *tw_slice, int64_t t_lo, int64_t t_hi, ShRec *out, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	int sidx = 0;
	for (int q = 1; q < off.n; ++q) sidx += off.b[q] <= i ? 1 : 0;
	ShRec r = req[i];
	if (r.kb < t_lo || r.kb >= t_hi) { atomicAdd(bad, 1ull); r.kb = -1; }
	else r.kb = (int64_t)tw_slice[r.kb - t_lo];
	out[(int64_t)sidx * stride + (i - off.b[sidx])] = r;
}

/* the answers -> merged positions inside the interval and the rows' symbols: reply.kb = row << 3 | symbol */
__global__ void __launch_bounds__(256) k_sh_place_text(int64_t n, const ShRec *rep, int64_t jlo, int64_t iv_start, int64_t *pos, uint8_t *b2rows, unsigned long long *bad)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const ShRec r = rep[i];
	const int64_t kb = (r.kb >> 3) - jlo;
	if (r.kb < 0 || kb < 0 || kb >= n || r.ka < iv_start) { atomicAdd(bad, 1ull); return; }
	pos[kb] = r.ka - iv_start + kb;
	b2rows[kb] = (uint8_t)(r.kb & 7);
}

/* ----------------------------------------------------------------------------------------- */
/* export                                                                                      */
/* ----------------------------------------------------------------------------------------- */

__global__ void __launch_bounds__(256) k_export_plain(IdxView ix, int64_t beg, int64_t end, uint8_t *out)
{
	int64_t i = beg + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (; i < end; i += stride) out[i - beg] = (uint8_t)idx_sym(ix, i);
}

/* run export (rb3_enc_fmr2fmd's leaf iteration, fm-index.c:31-52): the maximal runs of groups [g0, g0 + ng) as
 * start << 3 | sym, in BWT order.  A position starts a run iff its symbol differs from the one before it (runs continue
 * across slots, groups and chunks).  Pass 1 (EMIT = false) counts the run starts per group into column 0 of an 8 x u32
 * record (the engine's scan works on those), pass 2 writes them at the scanned offsets.
 * One wave per GROUP walks the group's slots -- a run slot IS its runs (one lane per code,
 * start = prefix sum of the lengths), a bit-plane slot is expanded as above -- so the cost follows the size of the block
 * array, not the number of symbols (a pangenome index: ~100x fewer codes than symbols).  A run that continues from the
 * slot (or group) before is recognised by the last symbol of that slot.  (Round 1 regenerated every symbol, one wave per
 * window: 44 ms for the 1.33 G symbols of config 3.) */
__device__ __forceinline__ uint32_t slot_code(const uint32_t *sp, int q) // code q of a run slot (q < 48)
{
	const uint32_t word = sp[(q / 6) * 4 + 1 + (q % 6) / 2];
	return (q & 1) ? word >> 16 : word & 0xFFFFu;
}

__device__ __forceinline__ uint32_t slot_plane_sym(const uint32_t *sp, uint32_t off) // symbol at offset off (< 256) of a bit-plane slot
{
	const uint32_t *w = sp + (off >> 5) * 4, bit = off & 31;
	return ((w[1] >> bit) & 1u) | ((w[2] >> bit) & 1u) << 1 | ((w[3] >> bit) & 1u) << 2;
}

/* last symbol of a slot (all lanes of the wave get it) */
__device__ __forceinline__ uint32_t slot_last_sym(const uint32_t *sp, int lane)
{
	const uint32_t hdr0 = sp[0], nsym = sp[28] - 0u;
	if (!(hdr0 & RB3_SLOT_RLE)) return slot_plane_sym(sp, (nsym & 0xFFFFu) - 1u);
	const uint32_t code = lane < RB3_RLE_CODES ? slot_code(sp, lane) : 7u;
	const int nc = __popcll(__ballot((code & 7u) != 7u));
	return wave_read(code, nc > 0 ? nc - 1 : 0) & 7u;
}

template<bool EMIT>
__global__ void __launch_bounds__(256) k_export_runs_g(IdxView ix, int64_t g0, int64_t ng, uint32_t *cnt8, const uint64_t *off8, uint64_t *runs)
{
	const int lane = threadIdx.x & 63;
	const int64_t gi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (gi >= ng) return;
	const int64_t g = g0 + gi, P0 = g << RB3_GRP_BITS;
	uint32_t total = 0;
	if (P0 < ix.n) {
		const uint64_t sm = ix.grp64[g * 8 + 6];
		const uint32_t slot0 = (uint32_t)sm, nsl = (uint32_t)__popc((uint32_t)(sm >> 32));
		// headers that carry the whole LF base (RB3_ABS_HEADERS) only differ in words 1..6; words 0 and 7 are what is read here
		uint32_t prev = 8u;
		if (g > 0) prev = slot_last_sym((const uint32_t*)(ix.slot16 + (int64_t)(slot0 - 1u) * 8), lane);
		uint64_t o = EMIT ? off8[gi * 8] : 0;
		for (uint32_t si = 0; si < nsl; ++si) {
			const uint32_t *sp = (const uint32_t*)(ix.slot16 + (int64_t)(slot0 + si) * 8);
			const uint32_t hdr0 = sp[0];
			const int64_t start = P0 + (int64_t)(hdr0 & 0xFFFFu);
			if (hdr0 & RB3_SLOT_RLE) {
				const uint32_t code = lane < RB3_RLE_CODES ? slot_code(sp, lane) : 7u;
				const uint32_t sym = code & 7u;
				uint32_t ex = wave_up1(RB3_RUN_END(code)); // cumulative codes: a run starts where the code before it ends
				if (lane == 0) ex = 0u;
				uint32_t before = wave_up1(sym);
				if (lane == 0) before = prev;
				const uint64_t H = __ballot(sym != 7u && sym != before);
				if (EMIT && (H >> lane & 1ull)) runs[o + __popcll(H & ((1ull << lane) - 1ull))] = (uint64_t)(start + ex) << 3 | sym;
				const int nh = __popcll(H), nc = __popcll(__ballot(sym != 7u));
				o += nh, total += (uint32_t)nh;
				if (nc > 0) prev = wave_read(sym, nc - 1);
			} else {
				const uint32_t nsym = sp[28] & 0xFFFFu; // (<= 256)
#pragma unroll
				for (int u = 0; u < 4; ++u) {
					const uint32_t off = 64u * u + lane;
					const uint32_t sym = off < nsym ? slot_plane_sym(sp, off) : 7u;
					uint32_t before = wave_up1(sym);
					if (lane == 0) before = prev;
					const uint64_t H = __ballot(sym != 7u && sym != before);
					if (EMIT && (H >> lane & 1ull)) runs[o + __popcll(H & ((1ull << lane) - 1ull))] = (uint64_t)(start + off) << 3 | sym;
					const int nh = __popcll(H);
					o += nh, total += (uint32_t)nh;
					const int nv = (int)nsym - 64 * u;
					if (nv > 0) prev = wave_read(sym, nv >= 64 ? 63 : nv - 1);
				}
			}
		}
	}
	if (!EMIT && lane < 8) cnt8[gi * 8 + lane] = lane == 0 ? total : 0u;
}

/* ----------------------------------------------------------------------------------------- */
/* sampled suffix array (rb3_ssa_gen, ssa.c:17-81)                                             */
/* ----------------------------------------------------------------------------------------- */

/* The reference walks LF from every sentinel row k0 to the start of string k0, one thread per
 * string (kt_for over sa->m, ssa.c:73), and notes text offset and string id at every row k with
 * (k - m) a multiple of 2^ss.  Here the rows are cut into sublists by SPLITTERS -- the m sentinel
 * rows (heads of the strings) and every row k >= m with (k - m) a multiple of 2^S -- and every
 * splitter walks its sublist at the same time (k_ssa_walk), leaving for each sampled row the pair
 * (splitter, steps from it).  One thread per string then hops from splitter to splitter
 * (k_ssa_link: 2^-S of the steps) to find where each sublist starts in its string, and a streaming
 * pass (k_ssa_final) turns the pairs into the reference's words.  Rows are in suffix order, so
 * sublist lengths are geometric with mean 2^S whatever the text looks like. */

#define RB3_SSA_LBITS 24              /* bits for the steps inside one sublist */
#ifndef RB3_SSA_END
#define RB3_SSA_END   (1ull << 63)    /* nxt word: the sublist ends at the start of its string */
#endif

/* symbol at offset `off` of the slot, this lane's share: 8 | symbol in the lane that holds it, else 0 */
__device__ __forceinline__ uint32_t slice_sym(const uint4 &sl, uint32_t hdr0, uint32_t off, int j)
{
	if (!(hdr0 & RB3_SLOT_RLE)) {
		const int t = (int)off - 32 * j;
		if (t < 0 || t >= 32) return 0u;
		return 8u | ((sl.y >> t) & 1u) | ((sl.z >> t) & 1u) << 1 | ((sl.w >> t) & 1u) << 2;
	} else {
		const uint32_t e[6] = { sl.y & 0xFFFFu, sl.y >> 16, sl.z & 0xFFFFu, sl.z >> 16, sl.w & 0xFFFFu, sl.w >> 16 };
		uint32_t pos = oct_prev_end(pk_ends(sl.w), j) >> 16, r = 0u; // cumulative codes: run i covers [end of the code before it, its own end)
#pragma unroll
		for (int i = 0; i < 6; ++i) {
			const uint32_t en = RB3_RUN_END(e[i]);
			if (off >= pos && off < en && (e[i] & 7u) != 7u) r = 8u | (e[i] & 7u);
			pos = en;
		}
		return r;
	}
}

/* one LF step of the index on itself: *c = B[k], returns C[c] + rank(c, k) (fm-index.h:109-112 + ssa.c:26-27) */
__device__ __forceinline__ int64_t oct_lf_self(const IdxView &ix, int64_t k, int j, int *c)
{
	RankLoad r;
	oct_rank_issue(ix, k, j, r);
	const uint32_t hdr0 = oct_bcast0(r.sl.x, j);
	const uint32_t off = r.koff - (hdr0 & 0xFFFFu);
	const uint32_t sy = oct_sum(slice_sym(r.sl, hdr0, off, j)) & 7u;
	*c = (int)sy;
	return oct_rank_finish(r, (int)sy, j, ix.abs);
}

/* nxt[2p] = next splitter (or RB3_SSA_END | sentinel row reached), nxt[2p+1] = steps of the sublist;
 * ssa[x] = p << 24 | steps for every sampled row on it; err[0] counts sublists too long to encode */
__global__ void __launch_bounds__(256) k_ssa_walk(IdxView ix, int S, int ss, int64_t nsp, uint64_t *nxt, uint64_t *ssa, unsigned long long *qhead, unsigned long long *err)
{
	const int lane = threadIdx.x & 63, j = lane & 7;
	const int64_t m = ix.m, maskS = (1LL << S) - 1, maskss = (1LL << ss) - 1;
	bool active = false;
	int64_t k = 0, p = 0;
	uint32_t l = 0;
	for (;;) {
		if (!active) {
			uint32_t w0 = 0, w1 = 0;
			if (j == 0) {
				unsigned long long w = atomicAdd(qhead, 1ull);
				w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
			}
			w0 = oct_bcast0(w0, j), w1 = oct_bcast0(w1, j);
			p = (int64_t)((uint64_t)w1 << 32 | w0);
			if (p >= nsp) break;
			k = p < m ? p : m + ((p - m) << S);
			l = 0, active = true;
		}
		do {
			int c;
			const int64_t k2 = oct_lf_self(ix, k, j, &c);
			++l;
			const int64_t km = k2 - m; // >= 0 whenever c != 0
			if (c != 0 && (km & maskss) == 0 && j == 0) ssa[km >> ss] = (uint64_t)p << RB3_SSA_LBITS | l;
			const bool at_split = c != 0 && (km & maskS) == 0;
			const bool too_long = l >= (1u << RB3_SSA_LBITS) - 1u;
			if (c == 0 || at_split || too_long) {
				if (j == 0) {
					nxt[2 * p] = c == 0 ? (RB3_SSA_END | (uint64_t)k2) : (uint64_t)(m + (km >> S));
					nxt[2 * p + 1] = l;
					if (too_long && c != 0 && !at_split) atomicAdd(err, 1ull);
				}
				active = false;
			}
			k = k2;
		} while (__all(active));
	}
}

/* Pointer jumping over the splitters: after round r, lnk[2p] points 2^r splitters ahead (or holds
 * RB3_SSA_END | the sentinel row the string ends in) and lnk[2p+1] is the number of LF steps to get there;
 * ceil(log2(#splitters)) + 1 rounds make every entry (END | row, steps to the start of the string).
 * Double-buffered: reads `in`, writes `out`. */
__global__ void __launch_bounds__(256) k_ssa_jump(int64_t nsp, const uint64_t *in, uint64_t *out)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= nsp) return;
	const ulonglong2 a = ((const ulonglong2*)in)[p];
	ulonglong2 r = a;
	if (!(a.x & RB3_SSA_END) && a.x < (uint64_t)nsp) {
		const ulonglong2 q = ((const ulonglong2*)in)[a.x];
		r.x = q.x, r.y = a.y + q.y;
	}
	((ulonglong2*)out)[p] = r;
}

/* the heads of the strings (splitters 0..m-1 = the sentinel rows): r2i[row reached] = string (ssa.c:36) */
__global__ void __launch_bounds__(256) k_ssa_heads(int64_t m, const uint64_t *lnk, uint64_t *r2i, unsigned long long *err)
{
	const int64_t k0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k0 >= m) return;
	const uint64_t e = lnk[2 * k0];
	const uint64_t row = e & ~RB3_SSA_END;
	if ((e & RB3_SSA_END) && row < (uint64_t)m) r2i[row] = (uint64_t)k0;
	else atomicAdd(err + 1, 1ull); // a cycle or a broken link: cannot happen with a valid index
}

/* ssa[x] = (offset of the sampled row's suffix in its string) << ms | string  (ssa.c:38-39).  The row was
 * reached l steps after splitter p, which is D = lnk[2p+1] steps from the start of the string: the offset is
 * D - l - 1 (the walk's last step lands on the sentinel row of the previous string, not on a suffix) */
__global__ void __launch_bounds__(256) k_ssa_final(int64_t n_ssa, int ms, int64_t m, const uint64_t *lnk, const uint64_t *r2i, uint64_t *ssa, unsigned long long *err)
{
	const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= n_ssa) return;
	const uint64_t t = ssa[x], p = t >> RB3_SSA_LBITS, l = t & ((1ull << RB3_SSA_LBITS) - 1);
	const uint64_t e = lnk[2 * p], row = e & ~RB3_SSA_END;
	if (!(e & RB3_SSA_END) || row >= (uint64_t)m) { atomicAdd(err + 1, 1ull); return; }
	ssa[x] = (lnk[2 * p + 1] - l - 1) << ms | r2i[row];
}

#endif
