/*
 * rb3gpu_planes.h -- plane-space interleave + rebuild: one block per 8192-symbol GROUP of the new index,
 * one lane per 32 output symbols.
 *
 * What it restates: worker_mgins + rope_insert_run (fm-index.c:237-249, rope.c:114-148) -- merged[pos[kb]] = B2[kb], the
 * symbols of B1 keep their order -- for the groups whose content is best handled as SYMBOLS: an index of reads or of few
 * genomes (bit-plane slots), the first rounds of a pangenome build (bit-plane and run slots mixed), and whatever the
 * run-space rebuild (k_reb_group) hands on.  Rounds 1-3 regenerated such windows symbol by symbol, one wave per 256
 * symbols (k_pass1w: ~640 vector instructions per window, 2.5 per symbol); here a lane owns one 32-symbol WORD of the
 * three bit planes and the insertion of the batch rows is a funnel shift plus a 32-bit parallel deposit per plane:
 *
 *   old range   the old symbols of the group are consecutive positions [A0, A0 + 8192 - rows) of the old index: at most 258
 *               old words, copied (bit-plane slots) or expanded (run slots: binary search + fill per word) into LDS once
 *   rows        the batch rows of the group set a bit per row in a row mask R and in three planes N (LDS atomics, by the
 *               whole block: balanced whatever the distribution over the windows)
 *   word        out_p = pdep(old_p >> shift, ~R) | N_p with shift = (old offset of the word) & 31, old offset = output offset -
 *               rows before it (a popcount scan over the octet of the window); per-symbol counts, run heads and the
 *               symbol that ends the word come from plane logic
 *   slots       the partition rule of k_decide (largest aligned power-of-two blocks of windows with <= 48 runs) on the
 *               head counts; bit-plane slots are the planes + a header, run slots are cut out of the head list
 *
 * ~0.3 vector instructions per symbol instead of 2.5.  Slots go to a scratch of 32 slots per group (final places need the
 * scan over all groups); k_place_pg copies them.  LISTED: the groups of a list (what k_reb_group left), scratch by list place.
 */
#ifndef RB3GPU_PLANES_H
#define RB3GPU_PLANES_H

#define RB3_PG_MAXOLD 36 /* old slots the old range of one group can touch (33 windows of two groups), with room to spare */
#define RB3_PG_BIG    0x3FFFu /* "start" of an unused run code: behind every offset of a slot */

struct PgLds {
	unsigned long long rm[2][256];        // batch rows by output word: [0] = row mask | plane 0 << 32, [1] = plane 1 | plane 2 << 32
	uint32_t opl[3][264];                 // planes of the old symbols by old word (k - k0)
	uint32_t rtab[RB3_PG_MAXOLD * 48];    // run slots of the old range: start << 3 | sym per code; afterwards the heads of the new run slots
	int32_t srel[RB3_PG_MAXOLD];          // old slot q: old offset of its first symbol - 32 k0 (may be negative)
	uint32_t skind[RB3_PG_MAXOLD];        // != 0: a run slot
	uint32_t top[256];                    // the symbol that ends every output word (8: none)
	uint32_t wcnt[3][34];                 // per window: symbol counts, two 16-bit fields per word; after the partition their exclusive prefix
	uint32_t wh[34];                      // run heads per window; after the partition their exclusive prefix
	uint32_t wfirst[32];                  // 1: the window's first symbol starts a run
	uint32_t part[4];                     // slot-start mask of the group
};

/* deposit the low bits of x at the set bits of m, in order (Hacker's Delight 7-5, "expand"), for three words that share the mask */
__device__ __forceinline__ void pdep3(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t m0)
{
	uint32_t mv[5], m = m0, mk = ~m << 1;
#pragma unroll
	for (int i = 0; i < 5; ++i) {
		uint32_t mp = mk ^ (mk << 1);
		mp ^= mp << 2, mp ^= mp << 4, mp ^= mp << 8, mp ^= mp << 16;
		mv[i] = mp & m;
		m = (m ^ mv[i]) | (mv[i] >> (1 << i));
		mk &= ~mp;
	}
#pragma unroll
	for (int i = 4; i >= 0; --i) {
		x0 = (x0 & ~mv[i]) | ((x0 << (1 << i)) & mv[i]);
		x1 = (x1 & ~mv[i]) | ((x1 << (1 << i)) & mv[i]);
		x2 = (x2 & ~mv[i]) | ((x2 << (1 << i)) & mv[i]);
	}
	x0 &= m0, x1 &= m0, x2 &= m0;
}

/* gstat[g*8 + 0..5] = symbol counts of the group, [6] = slots, [7] = slot-start mask (as k_decide / k_reb_group);
 * pslots[(u * 32 + i) * 8 + j]: slice j of the group's i-th slot, u = g (or the place in the list), headers relative to the group.
 * stat[0] += groups the run-space rebuild could have taken (no bit-plane slot and at most 13 slots in the old range, at most 8 new slots). */
template<bool LISTED>
__global__ void __launch_bounds__(256) k_plane_group(IdxView old, const int64_t *pos, const uint8_t *b2, int64_t n2, int64_t ntot, const int64_t *jw, int64_t nwin, int64_t ngrp,
		uint32_t *gstat, uint4 *pslots, const unsigned long long *skip, const uint32_t *glist, const uint32_t *nglist, uint32_t lcap, unsigned long long *lover, unsigned long long *stat, uint32_t *gpos = nullptr)
{
	__shared__ PgLds L;
	if (RB3_REB_SKIP(skip)) return;
	if (LISTED && *nglist > lcap) { // the list does not fit the scratch: tell the host (it emits the rebuild again with full sizes)
		if (blockIdx.x == 0 && threadIdx.x == 0) *lover = 1;
		return;
	}
	const int t = threadIdx.x, lane = t & 63, j = t & 7, lw = t >> 3;
	const int64_t nunits = LISTED ? (int64_t)*nglist : ngrp;
	L.rm[0][t] = 0ull, L.rm[1][t] = 0ull;
	for (int i = t; i < 264; i += 256) L.opl[0][i] = 0u, L.opl[1][i] = 0u, L.opl[2][i] = 0u;
	__syncthreads();
	for (int64_t u = blockIdx.x; u < nunits; u += gridDim.x) {
		const int64_t g = LISTED ? (int64_t)glist[u] : u;
		if (LISTED && gpos != nullptr && t == 0) gpos[g] = (uint32_t)u; // (where k_scan_place finds this group's slots)
		const int64_t P0 = g << RB3_GRP_BITS, wbase = g * RB3_GRP_WINS;
		const int64_t wend = wbase + RB3_GRP_WINS < nwin ? wbase + RB3_GRP_WINS : nwin;
		const int nvw = (int)(wend - wbase); // windows of the group that exist
		const int64_t j0 = jw[wbase], j1 = jw[wend];
		const int64_t jmine = jw[wbase + (lw < nvw ? lw : 0)];
		const int64_t rem = ntot - P0;
		const int nsg = rem >= RB3_GRP ? RB3_GRP : rem > 0 ? (int)rem : 0; // symbols of the group
		const int64_t M64 = j1 - j0, A0 = P0 - j0;
		bool valid = nvw > 0 && M64 >= 0 && M64 <= nsg && A0 >= 0 && A0 + (nsg - M64) <= old.n; // (anything else: an invalid pos[], caught by the validation)
		const int M = valid ? (int)M64 : 0, nold = valid ? nsg - M : 0;
		const int64_t k0 = A0 >> 5;
		const int base5 = (int)(A0 & 31);
		const int nw = nold > 0 ? (int)(((A0 + nold - 1) >> 5) - k0) + 1 : 0; // old words of the range (<= 258)
		// ---- phase 0: the first rows and the old slots are requested (row masks and old planes are clear: see below) ----
		int64_t rp = 0;
		uint32_t rs = 0;
		if (t < M) rp = pos[j0 + t], rs = b2[j0 + t];
		uint64_t sma = 0, smb = 0;
		int64_t ga = 0, gb = 0, fa = 0;
		uint32_t slot0b = 0;
		int ns = 0;
		if (nold > 0) {
			ga = A0 >> RB3_GRP_BITS, gb = (A0 + nold - 1) >> RB3_GRP_BITS;
			sma = old.gsm[ga], smb = old.gsm[gb];
			const uint32_t wa = ((uint32_t)A0 & (RB3_GRP - 1)) >> RB3_WIN_BITS, wb = ((uint32_t)(A0 + nold - 1) & (RB3_GRP - 1)) >> RB3_WIN_BITS;
			fa = (int64_t)((uint32_t)sma + __popc((uint32_t)(sma >> 32) & ((2u << wa) - 1u)) - 1u);
			const int64_t la = (int64_t)((uint32_t)smb + __popc((uint32_t)(smb >> 32) & ((2u << wb) - 1u)) - 1u);
			slot0b = (uint32_t)smb;
			ns = (int)(la - fa + 1);
			if (ns < 0 || ns > RB3_PG_MAXOLD) ns = 0, valid = false; // (a directory that contradicts itself: never with a valid index)
		}
		// ---- phase A1: the old slots of the range, an octet each ----
		uint32_t anybp = 0;
		for (int q = lw; q < ns; q += 32) {
			const int64_t s = fa + q;
			const uint4 sl = old.slot16[s * 8 + j];
			const uint32_t hdr0 = oct_bcast0(sl.x, j);
			const int64_t sgrp = (gb != ga && (uint32_t)s >= slot0b) ? gb : ga;
			const int rel = (int)(((sgrp << RB3_GRP_BITS) + (int64_t)(hdr0 & 0xFFFFu)) - (k0 << 5));
			if (!(hdr0 & RB3_SLOT_RLE)) {
				const int wi = (rel >> 5) + j;
				if (wi >= 0 && wi < 264) L.opl[0][wi] = sl.y, L.opl[1][wi] = sl.z, L.opl[2][wi] = sl.w;
				anybp = 1;
			} else {
				const uint32_t e[6] = { sl.y & 0xFFFFu, sl.y >> 16, sl.z & 0xFFFFu, sl.z >> 16, sl.w & 0xFFFFu, sl.w >> 16 };
				// cumulative codes: a run starts where the code before it ends (the first one of a lane: where the lane below ended)
				uint32_t st = oct_prev_end(pk_ends(sl.w), j) >> 16;
#pragma unroll
				for (int i = 0; i < 6; ++i) {
					const uint32_t en = RB3_RUN_END(e[i]);
					L.rtab[q * 48 + j * 6 + i] = ((e[i] & 7u) != 7u && en > st ? st : RB3_PG_BIG) << 3 | (e[i] & 7u);
					st = en;
				}
			}
			if (j == 0) L.srel[q] = rel, L.skind[q] = hdr0 & RB3_SLOT_RLE;
		}
		if (stat != nullptr) { // (could the run-space rebuild take this group?  no bit-plane slot and few enough slots in the old range, and -- below -- few new slots)
			const bool bp = __syncthreads_or((int)anybp) != 0;
			if (t == 0) L.part[1] = !bp && nold > 0 && ns * RB3_RLE_CODES <= 624 ? 1u : 0u;
		} else
			__syncthreads();
		// ---- rows: a bit per batch row in the row mask and the three planes of its output word ----
		for (int r = t; r < M; r += 256) {
			if (r >= 256) rp = pos[j0 + r], rs = b2[j0 + r];
			const int64_t pp = rp - P0;
			if (pp >= 0 && pp < RB3_GRP) {
				const uint32_t bit = 1u << ((uint32_t)pp & 31u);
				atomicOr(&L.rm[0][pp >> 5], (unsigned long long)bit | ((rs & 1u) ? (unsigned long long)bit << 32 : 0ull));
				if (rs & 6u) atomicOr(&L.rm[1][pp >> 5], ((rs & 2u) ? (unsigned long long)bit : 0ull) | ((rs & 4u) ? (unsigned long long)bit << 32 : 0ull));
			}
		}
		// ---- phase A2: the old words that lie in run slots, a lane each ----
		for (int idx = t; idx < nw; idx += 256) {
			const int64_t ow = (k0 + idx) >> 3; // old window
			const uint64_t sm = (ow >> 5) == ga ? sma : smb;
			const int q = (int)((int64_t)((uint32_t)sm + __popc((uint32_t)(sm >> 32) & ((2u << ((uint32_t)ow & 31u)) - 1u)) - 1u) - fa);
			if (q < 0 || q >= ns || !L.skind[q]) continue;
			const uint32_t off = (uint32_t)(idx * 32 - L.srel[q]);
			const uint32_t *rt = &L.rtab[q * 48];
			int r = 0;
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1)
				if (r + d < RB3_RLE_CODES && (rt[r + d] >> 3) <= off) r += d;
			uint32_t p0 = 0, p1 = 0, p2 = 0, at = 0;
			for (; at < 32u && r < RB3_RLE_CODES; ++r) {
				const uint32_t e = rt[r], nxt = r + 1 < RB3_RLE_CODES ? rt[r + 1] >> 3 : RB3_PG_BIG;
				uint32_t end = nxt - off;
				end = nxt <= off ? at : end > 32u ? 32u : end; // (nxt <= off only with a corrupt slot)
				const uint32_t m = (end >= 32u ? 0xFFFFFFFFu : (1u << end) - 1u) & ~((1u << at) - 1u);
				if (e & 1u) p0 |= m;
				if (e & 2u) p1 |= m;
				if (e & 4u) p2 |= m;
				at = end > at ? end : 32u;
			}
			L.opl[0][idx] = p0, L.opl[1][idx] = p1, L.opl[2][idx] = p2;
		}
		__syncthreads();
		// ---- phase B1: the output word of this lane ----
		const uint32_t R = (uint32_t)L.rm[0][t];
		uint32_t o0 = (uint32_t)(L.rm[0][t] >> 32), o1 = (uint32_t)L.rm[1][t], o2 = (uint32_t)(L.rm[1][t] >> 32);
		const uint32_t pc = __popc(R);
		const int cbefore = (int)oct_exscan(pc, j);
		const int wofs = lw * RB3_WIN + 32 * j; // offset of the word in the group
		const int nval = nsg - wofs >= 32 ? 32 : nsg - wofs > 0 ? nsg - wofs : 0;
		const uint32_t vmask = nval >= 32 ? 0xFFFFFFFFu : (1u << nval) - 1u;
		{
			int oo = base5 + wofs - (int)(jmine - j0) - cbefore; // old offset of the word's first old symbol, from 32 k0
			oo = oo < 0 ? 0 : oo > 262 * 32 ? 262 * 32 : oo;     // (only an invalid pos[] is outside)
			const int idx = oo >> 5, sh = oo & 31;
			uint32_t x0 = __builtin_amdgcn_alignbit(L.opl[0][idx + 1], L.opl[0][idx], sh);
			uint32_t x1 = __builtin_amdgcn_alignbit(L.opl[1][idx + 1], L.opl[1][idx], sh);
			uint32_t x2 = __builtin_amdgcn_alignbit(L.opl[2][idx + 1], L.opl[2][idx], sh);
			pdep3(x0, x1, x2, ~R & vmask);
			o0 = (o0 & R) | x0 | ~vmask, o1 = (o1 & R) | x1 | ~vmask, o2 = (o2 & R) | x2 | ~vmask; // (positions past the end: 7)
		}
		{
			const uint32_t n0 = ~o0, n1 = ~o1, n2m = ~o2 & vmask;
			const uint32_t a00 = n2m & n1, a01 = n2m & o1, a10 = o2 & n1 & vmask;
			const uint32_t c01 = __popc(a00 & n0) | __popc(a00 & o0) << 16, c23 = __popc(a01 & n0) | __popc(a01 & o0) << 16, c45 = __popc(a10 & n0) | __popc(a10 & o0) << 16;
			const uint32_t s01 = oct_sum(c01), s23 = oct_sum(c23), s45 = oct_sum(c45);
			if (j == 0) L.wcnt[0][lw] = s01, L.wcnt[1][lw] = s23, L.wcnt[2][lw] = s45;
		}
		L.top[t] = nval > 0 ? ((o0 >> (nval - 1)) & 1u) | ((o1 >> (nval - 1)) & 1u) << 1 | ((o2 >> (nval - 1)) & 1u) << 2 : 8u;
		__syncthreads();
		// (row masks and old planes have been read: cleared for the next group of this block, whose first writes to them come behind two more barriers)
		L.rm[0][t] = 0ull, L.rm[1][t] = 0ull;
		for (int i = t; i < 264; i += 256) L.opl[0][i] = 0u, L.opl[1][i] = 0u, L.opl[2][i] = 0u;
		// ---- phase B2: run heads (a symbol that differs from the one before it, across words and windows) ----
		uint32_t G;
		{
			const uint32_t prev = t > 0 ? L.top[t - 1] : 0u;
			G = ((o0 ^ (o0 << 1 | (prev & 1u))) | (o1 ^ (o1 << 1 | (prev >> 1 & 1u))) | (o2 ^ (o2 << 1 | (prev >> 2 & 1u)))) & vmask;
			if (t == 0) G |= vmask & 1u; // the group's first symbol
		}
		const uint32_t hc = __popc(G);
		{
			const uint32_t hs = oct_sum(hc);
			if (j == 0) L.wh[lw] = lw < nvw ? hs : 0u, L.wfirst[lw] = G & 1u;
		}
		__syncthreads();
		// ---- the slot partition (k_decide's rule), one wave ----
		if (t < 64) {
			const uint32_t v = lane < RB3_GRP_WINS ? L.wh[lane] : 0u;
			const uint32_t inc = wave_incl_scan(v);
			uint32_t c0 = lane < RB3_GRP_WINS ? L.wcnt[0][lane] : 0u, c1 = lane < RB3_GRP_WINS ? L.wcnt[1][lane] : 0u, c2 = lane < RB3_GRP_WINS ? L.wcnt[2][lane] : 0u;
			const uint32_t i0 = wave_incl_scan(c0), i1 = wave_incl_scan(c1), i2 = wave_incl_scan(c2);
			wave_sync();
			if (lane <= RB3_GRP_WINS) L.wh[lane] = inc - v, L.wcnt[0][lane] = i0 - c0, L.wcnt[1][lane] = i1 - c1, L.wcnt[2][lane] = i2 - c2;
			wave_sync();
			int level = 0;
			if (lane < RB3_GRP_WINS) {
#pragma unroll
				for (int jl = 1; jl <= 5; ++jl) {
					const int sz = 1 << jl, a = lane & ~(sz - 1);
					const int runs = a + sz <= nvw ? (int)L.wh[a + sz] - (int)L.wh[a] + 1 - (int)L.wfirst[a] : 1 << 20;
					if (runs <= RB3_RLE_CODES && level == jl - 1) level = jl;
				}
			}
			const bool sstart = lane < nvw && (lane & ((1 << level) - 1)) == 0;
			const uint32_t mask = (uint32_t)__ballot(sstart);
			if (lane == 0) {
				L.part[0] = mask;
				if (stat != nullptr && L.part[1] && __popc(mask) <= 8) atomicAdd(stat, 1ull); // (the host picks the next merge's rebuild by this count)
			}
			if (lane < 8) {
				const uint32_t tot01 = wave_read(i0, 63), tot23 = wave_read(i1, 63), tot45 = wave_read(i2, 63);
				const uint32_t vv = lane == 0 ? (tot01 & 0xFFFFu) : lane == 1 ? tot01 >> 16 : lane == 2 ? (tot23 & 0xFFFFu) : lane == 3 ? tot23 >> 16 :
					lane == 4 ? (tot45 & 0xFFFFu) : lane == 5 ? tot45 >> 16 : lane == 6 ? (uint32_t)__popc(mask) : mask;
				gstat[g * 8 + lane] = valid ? vv : 0u;
			}
		}
		__syncthreads();
		// ---- slots: bit-plane slots at once, run slots through the list of their heads ----
		const uint32_t mask = L.part[0];
		uint32_t *hl = L.rtab;
		int a = 0, an = 0, si = 0, nr = 0;
		uint32_t hq = 0;
		const bool mine = lw < nvw && valid;
		if (mine) {
			const uint32_t below = mask & ((2u << lw) - 1u);
			a = 31 - __clz((int)below), si = __popc(below) - 1;
			const uint32_t nx = a == 31 ? 0u : mask & ~((2u << a) - 1u);
			an = nx ? __ffs((int)nx) - 1 : nvw;
			const int nsym = nsg - a * RB3_WIN >= (an - a) * RB3_WIN ? (an - a) * RB3_WIN : nsg - a * RB3_WIN > 0 ? nsg - a * RB3_WIN : 0;
			hq = j == 0 ? (uint32_t)(a * RB3_WIN) | (an - a > 1 ? RB3_SLOT_RLE : 0u) : j == 7 ? (uint32_t)nsym :
				(L.wcnt[(j - 1) >> 1][a] >> (((j - 1) & 1) * 16)) & 0xFFFFu;
			nr = (int)L.wh[an] - (int)L.wh[a] + 1 - (int)L.wfirst[a];
			if (an - a == 1) pslots[((int64_t)u * RB3_GRP_WINS + si) * 8 + j] = make_uint4(hq, o0, o1, o2);
			else {
				int rk = si * 49 + (int)L.wh[lw] - (int)L.wh[a] + 1 - (int)L.wfirst[a] + (int)oct_exscan(hc, j);
				for (uint32_t gm = G; gm; gm &= gm - 1u) {
					const int b = __ffs((int)gm) - 1;
					hl[rk++] = (uint32_t)(wofs + b) << 3 | ((o0 >> b) & 1u) | ((o1 >> b) & 1u) << 1 | ((o2 >> b) & 1u) << 2;
				}
				if (lw == a && j == 0) {
					if (!(G & 1u)) hl[si * 49] = (uint32_t)(a * RB3_WIN) << 3 | (o0 & 1u) | (o1 & 1u) << 1 | (o2 & 1u) << 2; // the run that comes in from the slot before
					hl[si * 49 + nr] = (uint32_t)(a * RB3_WIN + nsym) << 3 | 7u;
				}
			}
		}
		__syncthreads();
		if (mine && an - a > 1 && lw == a) { // a run slot: this octet writes it, six codes per lane
			uint32_t c[6];
			const uint32_t x0 = (uint32_t)(a * RB3_WIN), xe = hl[si * 49 + nr] >> 3; // the slot's first offset in the group, and where its symbols end
#pragma unroll
			for (int i = 0; i < 6; ++i) { // cumulative codes: where the run ends in the slot; the codes behind the last run repeat its end
				const int r = j * 6 + i;
				c[i] = RB3_RUN_CODE(xe > x0 ? xe - x0 : 1u, 7u);
				if (r < nr) {
					const uint32_t e0 = hl[si * 49 + r], e1 = hl[si * 49 + r + 1];
					c[i] = RB3_RUN_CODE((e1 >> 3) - x0, e0 & 7u);
				}
			}
			pslots[((int64_t)u * RB3_GRP_WINS + si) * 8 + j] = make_uint4(hq, c[0] | c[1] << 16, c[2] | c[3] << 16, c[4] | c[5] << 16);
		}
		__syncthreads(); // (the next group of this block clears what this one read)
	}
}

/* the slots k_plane_group left in its scratch, to their final places; the directory entry of the group.  One wave per group. */
template<bool LISTED>
__global__ void __launch_bounds__(256) k_place_pg(const uint32_t *gstat, const uint64_t *gpre, const uint64_t *tot, const uint4 *pslots, rb3_grp_t *grp, uint4 *slot16, int64_t ngrp, int64_t ntot,
		const unsigned long long *skip, const uint32_t *glist, const uint32_t *nglist, uint32_t lcap, uint64_t slot_cap, int64_t abs_lim)
{
	if (RB3_REB_SKIP(skip)) return;
	if (LISTED && *nglist > lcap) return;
	if (tot[6] > slot_cap) return; // (the slot array was sized by an estimate: the host looks at the total and emits again)
	const int lane = threadIdx.x & 63, j = lane & 7;
	const int64_t nunits = LISTED ? (int64_t)*nglist : ngrp;
	for (int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < nunits; u += ((int64_t)gridDim.x * blockDim.x) >> 6) {
		const int64_t g = LISTED ? (int64_t)glist[u] : u;
		const uint32_t gs = gstat[g * 8 + j];
		const uint64_t gp = gpre[g * 8 + j];
		uint64_t cb = 0;
		for (int a = 0; a < j && a < 6; ++a) cb += tot[a];
		const uint32_t ns = (uint32_t)__shfl((int)gs, 6, 8), mask = (uint32_t)__shfl((int)gs, 7, 8);
		const uint64_t slot0 = (uint64_t)(uint32_t)__shfl((int)(uint32_t)gp, 6, 8) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(gp >> 32), 6, 8) << 32;
		if (lane < 8) ((uint64_t*)grp)[g * 8 + j] = j < 6 ? cb + gp : j == 6 ? (uint64_t)(uint32_t)slot0 | (uint64_t)mask << 32 : 0ull;
		const uint32_t below = (uint32_t)__shfl((int)(uint32_t)(cb + gp), (j + 7) & 7, 8);
		const uint32_t add = (RB3_ABS_HEADERS(ntot, abs_lim) && j >= 1 && j <= 6) ? below : 0u;
		for (uint32_t si = (uint32_t)lane >> 3; si < ns && si < RB3_GRP_WINS; si += 8) {
			uint4 v = pslots[((int64_t)u * RB3_GRP_WINS + si) * 8 + j];
			v.x += add;
			slot16[(slot0 + si) * 8 + j] = v;
		}
	}
}

/* ----------------------------------------------------------------------------------------- */
/* the scan over the groups and the placement of their slots, in ONE kernel                     */
/* ----------------------------------------------------------------------------------------- */

/* Rounds 1-3 ran three scan kernels over the per-group records (gstat: six symbol counts + slot count), then k_place /
 * k_place_pg copied every slot from the scratch to its final position (fixing the headers on the way), then k_grp_compact
 * copied the directory's slot words: six launches, ~80 us per round of a 1.3 G-symbol build.  Here a block takes 256 groups:
 * exclusive scan of their records inside the block, the block's totals published, the totals of the blocks before it fetched
 * by a decoupled look-back (Merrill & Garland), and then the same lanes place the slots of those 256 groups -- 8 lanes per
 * group -- and write directory entry and compact slot word.  One launch; gstat is read once and the prefix never leaves the chip.
 *
 * lb: look-back state, 16 x u64 per block: [0..6] the block's totals, [8..14] its inclusive prefix, every word tagged with the
 * launch's epoch (never 0) in its upper 20 bits: the tag stands in for clearing and for memory ordering (see below).
 * C[] of the new index is known before the scan: acc of the old index + the batch's symbol counts (tot2, from the LF histogram). */
#ifndef RB3_SP_GROUPS
#define RB3_SP_GROUPS 256   /* groups per block (scanned by as many threads) */
#endif
#ifndef RB3_SP_THREADS
#define RB3_SP_THREADS 1024 /* threads per block (all of them place slots) */
#endif
struct SpLds {
	unsigned long long pre[8][RB3_SP_GROUPS]; // exclusive prefix of every group of the block (columns 0..5 symbols, 6 slots)
	uint32_t ns[RB3_SP_GROUPS], mask[RB3_SP_GROUPS];
	uint32_t wtot[4][8];
	unsigned long long bpre[8];
	uint32_t bid;
};

__global__ void __launch_bounds__(RB3_SP_THREADS) k_scan_place(const uint32_t *gstat, int64_t ngrp, int64_t ntot, const uint8_t *gkind, const uint4 *gslots, const uint4 *pslots, const uint32_t *gpos,
		rb3_grp_t *grp, uint64_t *gsm, uint4 *slot16, unsigned long long *dtot, unsigned long long *lb, unsigned int *ticket, unsigned long long epoch,
		Acc7 acc_old, const uint64_t *tot2, const unsigned long long *skip, const uint32_t *nglist, uint32_t lcap, uint64_t slot_cap, int64_t abs_lim)
{
	__shared__ SpLds L;
	if (RB3_REB_SKIP(skip)) return;
	if (nglist != nullptr && *nglist > lcap) return; // (the hand-over list did not fit its scratch: the host emits the rebuild again)
	const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
	const int64_t nblk = (ngrp + RB3_SP_GROUPS - 1) / RB3_SP_GROUPS;
	if (t == 0) {
		const unsigned int b = atomicAdd(ticket, 1u); // blocks take their numbers in the order they start: whoever a block waits for is running or done
		if ((int64_t)b + 1 == nblk) *ticket = 0u;     // (the last number of this launch: the next launch starts from 0 again)
		L.bid = b;
	}
	__syncthreads();
	const int64_t bid = L.bid;
	if (bid >= nblk) return;
	// ---- the records of this block's groups and their scan: the first RB3_SP_GROUPS threads, a group each ----
	uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, inc[7] = {0, 0, 0, 0, 0, 0, 0};
	const int64_t g = bid * RB3_SP_GROUPS + t;
	if (t < RB3_SP_GROUPS) {
		if (g < ngrp) {
			const uint4 a = ((const uint4*)gstat)[g * 2], b = ((const uint4*)gstat)[g * 2 + 1];
			v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
		}
#pragma unroll
		for (int c = 0; c < 7; ++c) inc[c] = wave_incl_scan(v[c]);
		if (lane == 63) {
#pragma unroll
			for (int c = 0; c < 7; ++c) L.wtot[wv][c] = inc[c];
		}
		L.ns[t] = v[6], L.mask[t] = v[7];
	}
	__syncthreads();
	uint32_t wbase[7], btot[7];
#pragma unroll
	for (int c = 0; c < 7; ++c) {
		wbase[c] = 0, btot[c] = 0;
		for (int w = 0; w < RB3_SP_GROUPS / 64; ++w) { const uint32_t x = L.wtot[w][c]; btot[c] += x; if (w < wv) wbase[c] += x; }
	}
	// ---- publish the block's totals, fetch the prefix of the blocks before it ----
	// Every published word carries the launch's tag in its upper 20 bits (value: 44 bits), is written with a relaxed agent-scope atomic
	// (written through) and read with one: no release fence -- on this chip a release at agent scope writes the XCD's whole L2 back, and
	// the L2 is full of the slots the neighbouring blocks are placing (measured: the kernel took 180 us with fences) -- and no ordering
	// between the words is needed, because a reader waits for the tag of EVERY word it uses.
	unsigned long long *my = lb + bid * 16; // [0..6] totals of the block, [8..14] inclusive prefix
	const unsigned long long tag = (epoch & 0xFFFFFull) << 44, vmask = (1ull << 44) - 1ull;
	if (t < 7) __hip_atomic_store(&my[t], tag | (unsigned long long)btot[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (t < 64) { // the look-back, 64 blocks per step (a lane each)
		unsigned long long p[7] = {0, 0, 0, 0, 0, 0, 0};
		for (int64_t base = bid - 1; base >= 0; base -= 64) {
			const int64_t q = base - lane;
			const bool valid = q >= 0;
			const unsigned long long *o = lb + (valid ? q : 0) * 16;
			unsigned long long xi = 0, xa = 0;
			bool isinc = false;
			if (valid)
				for (;;) { // column 6 of the inclusive prefix, else of the totals: whichever is there
					xi = __hip_atomic_load(&o[14], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((xi & ~vmask) == tag) { isinc = true; break; }
					xa = __hip_atomic_load(&o[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if ((xa & ~vmask) == tag) break;
				}
			const unsigned long long im = __ballot(isinc);
			const int first = im ? __ffsll((unsigned long long)im) - 1 : 64; // the nearest block that already knows its inclusive prefix
#pragma unroll
			for (int c = 0; c < 7; ++c) {
				unsigned long long x = 0;
				if (valid && lane <= first) { // (lanes before `first`: totals; lane `first`: the inclusive prefix)
					const unsigned long long *w = &o[(lane == first ? 8 : 0) + c];
					do { x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((x & ~vmask) != tag);
					x &= vmask;
				}
				// (sums of 64 totals stay below 2^32; the inclusive prefix needs both halves)
				const uint32_t lo = wave_read(wave_incl_scan((uint32_t)(lane < first ? x : 0ull)), 63);
				p[c] += lo;
				if (first < 64) p[c] += (unsigned long long)wave_read((uint32_t)x, first) | (unsigned long long)wave_read((uint32_t)(x >> 32), first) << 32;
			}
			if (first < 64) break;
		}
		if (t == 0) {
#pragma unroll
			for (int c = 0; c < 7; ++c) {
				L.bpre[c] = p[c];
				__hip_atomic_store(&my[8 + c], tag | ((p[c] + btot[c]) & vmask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (bid + 1 == nblk) { // the totals of the new index, for the host
#pragma unroll
				for (int c = 0; c < 7; ++c) dtot[c] = p[c] + btot[c];
				dtot[7] = 0ull;
			}
		}
	}
	__syncthreads();
	if (t < RB3_SP_GROUPS) {
#pragma unroll
		for (int c = 0; c < 7; ++c) L.pre[c][t] = L.bpre[c] + wbase[c] + (inc[c] - v[c]);
	}
	__syncthreads();
	// ---- placement: 8 lanes per group, all threads of the block; the slots of a group are asked for eight at a time ----
	const int j = t & 7;
	uint64_t cb = (uint64_t)acc_old.a[j < 6 ? j : 0]; // C[j] of the merged BWT (lanes 0..5): the old index's plus the batch symbols below j
	for (int a = 0; a < j && a < 6; ++a) cb += tot2[a];
	for (int k = t >> 3; k < RB3_SP_GROUPS; k += RB3_SP_THREADS / 8) {
		const int64_t gg = bid * RB3_SP_GROUPS + k;
		if (gg >= ngrp) break;
		const uint32_t ns = L.ns[k] < (uint32_t)RB3_GRP_WINS ? L.ns[k] : (uint32_t)RB3_GRP_WINS, mask = L.mask[k];
		const uint64_t slot0 = L.pre[6][k];
		const uint64_t gp = j < 6 ? L.pre[j][k] : 0ull;
		((uint64_t*)grp)[gg * 8 + j] = j < 6 ? cb + gp : j == 6 ? (uint64_t)(uint32_t)slot0 | (uint64_t)mask << 32 : 0ull;
		if (j == 6) gsm[gg] = (uint64_t)(uint32_t)slot0 | (uint64_t)mask << 32;
		const uint32_t below = (uint32_t)__shfl((int)(uint32_t)(cb + gp), (j + 7) & 7, 8); // header lane j holds symbol j - 1, whose base sits in lane j - 1
		const uint32_t add = (RB3_ABS_HEADERS(ntot, abs_lim) && j >= 1 && j <= 6) ? below : 0u;
		if (slot0 + ns > slot_cap) continue; // (the slot array was sized by an estimate: the host sees the total and emits again)
		const bool fromp = gkind == nullptr || gkind[gg] != 0;
		const uint4 *src = fromp ? pslots + (int64_t)(gpos ? gpos[gg] : (uint32_t)gg) * RB3_GRP_WINS * 8 : gslots + gg * RB3_RG_MAXSLOTS * 8;
		for (uint32_t s0 = 0; s0 < ns; s0 += 8) {
			uint4 x[8];
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) if (s0 + q < ns) x[q] = src[(s0 + q) * 8 + j];
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q)
				if (s0 + q < ns) {
					x[q].x += add;
					slot16[(slot0 + s0 + q) * 8 + j] = x[q];
				}
		}
	}
}

#endif
