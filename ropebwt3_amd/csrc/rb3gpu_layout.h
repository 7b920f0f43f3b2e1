/*
 * rb3gpu_layout.h -- the flat HBM block array that replaces the reference's
 * mrope/rope/rle B+-tree (mrope.h:10-14, rope.h:11-35, rle.h:36-75).
 *
 * The BWT of n symbols is cut into GROUPS of 8192 symbols and WINDOWS of 256
 * symbols (32 windows per group).  Storage is an array of 128-byte SLOTS plus
 * a 64-byte directory entry per group:
 *
 *   Grp  (64 B, index = position >> 13)
 *        u64 cnt[6]   C[a] + #{i < group start : B[i] = a}   (the C array is folded in,
 *                     so an LF step is one add: LF(a,k) = cnt[a] + in-group count)
 *        u32 slot0    index of the group's first slot
 *        u32 mask     bit w set  <=>  window w of the group starts a new slot
 *        u64 spare
 *   Slot (128 B) = 8 lane slices of 16 B: { u32 hdr; u32 w[3] }
 *        hdr[0]       bit 31: 1 = run slot, 0 = bit-plane slot; bits 0..15: offset of the
 *                     slot's first symbol from the group start (a multiple of 256)
 *        hdr[1..6]    #{group start <= i < slot start : B[i] = a}, a = 0..5 -- or, in a block array that
 *                     consists of bit-plane slots only and holds < 2^32 symbols (RB3_ABS_HEADERS in
 *                     rb3gpu_kernels.h), the whole LF base C[a] + #{i < slot start : B[i] = a}, so that
 *                     rank is a single memory request there
 *        hdr[7]       number of symbols covered by the slot
 *      bit-plane slot (exactly one window): slice j holds symbols [32j, 32j+32) as three
 *        32-bit planes; symbol = bit0 | bit1<<1 | bit2<<2; padding past the end = 7
 *      run slot (2, 4, 8, 16 or 32 whole windows, at most 48 runs): slice j holds six
 *        16-bit codes (end-1)<<3 | sym, `end` = the offset from the slot start just past the run, 1..8192
 *        (CUMULATIVE: run i covers [end of code i-1, end of code i), the first run starts at 0 -- a lane of the
 *        octet that decodes the slot gets the starts of its six runs from its own words and one value of the lane
 *        below, where the lengths of rounds 1-5 needed a prefix sum over the octet); unused codes have sym = 7 and
 *        repeat the end of the last used one (empty runs): the last code of a slot always holds the slot's symbols
 *
 * A slot never crosses a group, a run slot covers an aligned power-of-two number of
 * windows, and a window with more than 48 runs (or any single window) is a bit-plane
 * slot, so every slot is one 128-byte line and the slot holding offset k is found from
 * the group entry alone: slot0 + popcount(mask & ((2 << (k>>8 & 31)) - 1)) - 1.
 * rank() is therefore two dependent memory round trips (group entry, slot), and eight
 * lanes decode one slot cooperatively (16 B per lane, one coalesced 128-B request).
 *
 * Memory: 0.5 B/symbol worst case (all bit-plane) + 1/128 B/symbol of directory;
 * down to 1/64 B/symbol where runs are long.
 */
#ifndef RB3GPU_LAYOUT_H
#define RB3GPU_LAYOUT_H

#include <stdint.h>

#define RB3_WIN_BITS   8
#define RB3_WIN        256
#define RB3_GRP_BITS   13
#define RB3_GRP        8192
#define RB3_GRP_WINS   32
#define RB3_RLE_CODES  48
#define RB3_RLE_MAXLEN 8192
#define RB3_SLOT_RLE   0x80000000u
/* a run code from the run's exclusive end offset in the slot (1..8192) and its symbol; the end of a code */
#define RB3_RUN_CODE(end, sym) ((((uint32_t)(end) - 1u) << 3) | (uint32_t)(sym))
#define RB3_RUN_END(code) ((((uint32_t)(code)) >> 3) + 1u)

typedef struct {
	uint64_t cnt[6];
	uint32_t slot0, mask;
	uint64_t spare;
} rb3_grp_t; /* 64 bytes */

typedef struct {
	uint32_t hdr, w[3];
} rb3_slice_t; /* 16 bytes */

typedef struct {
	rb3_slice_t s[8];
} rb3_slot_t; /* 128 bytes */

#endif
