// Shared by the phase launches of rnn_resident.hip (4 waves per workgroup) and rnn_w8.hip (8): the multi-problem argument blocks
// and the chunk-publishing producer that runs inside a forward launch.  Included inside each file's anonymous namespace.
#pragma once
constexpr int RM_MAX = 8, RM_XP_MAX = 2;
struct rnn_fwd_multi {
    mvae_rnn_fwd_args p[RM_MAX];
    mvae_xpand_args xp[RM_XP_MAX];
    int32_t base[RM_XP_MAX + RM_MAX + 1];      // xpand problems first, then the recurrences
    int32_t nx, n;
};
struct rnn_bwd_multi {
    mvae_rnn_bwd_args p[RM_MAX];
    int32_t base[RM_MAX + 1];
    int32_t n;
};
// out (R, N) bf16 in TILE16 = xs[r] * w[n] + bias[n] - or, with idx, the rows table[idx[r]] of a one-hot layer's lookup table
// (x*W + b already) - chunk after chunk of chunk_rows rows (a chunk is contiguous in TILE16); stored write-through, every
// wave publishes each chunk once (the consumer expects (waves per workgroup) * nb increments per chunk).
// A wave's unit of work is 16 rows x a quarter of the columns (TPS tiles of 16 columns; a tile is 512 contiguous bytes, lane =
// 4 columns of one row): the row's scalar / table index is ONE load per unit and there is no division anywhere.  The loop is
// software-pipelined around the one memory counter gfx950 has for loads AND stores (vmcnt, retired in issue order): a load
// issued behind write-through stores is not usable before those stores are acknowledged by memory, so the loads of
// unit i+1 (and the key of unit i+2) are issued BEFORE the stores of unit i and waited for with the stores still in flight.
// (history: one quad per thread and round with 64-bit index arithmetic moved 0.6 GB/s per workgroup, the unit loop with
//  load-then-store batches 2.2 GB/s - 16 workgroups then took 3.5 ms for the 32 MB the encoder's bottom layer reads)
template <int TPS, bool GATHER>
__device__ __forceinline__ void xpand_pipe(const mvae_xpand_args& x, const int wave, const int nw, const int lane) {
    const int N = x.N, ntn = N >> 4, nchunks = x.R / x.chunk_rows, rbs = x.chunk_rows >> 4, units = rbs << 2;
    const int col = (lane >> 4) * 4, tn0 = (wave & 3) * TPS;        // (u & 3 == wave & 3 for every unit of this wave: nw % 4 == 0)
    const size_t chunk_bytes = (size_t)x.chunk_rows * N * 2;
    if (wave >= units) {
        for (int c = 0; c < nchunks; ++c) wave_signal_done<false>(x.chunk_done + c);
        return;
    }
    const bf16_t* __restrict__ table = reinterpret_cast<const bf16_t*>(x.table) + tn0 * 16 + col;
    auto key_of = [&](int cc, int uu) -> unsigned {
        const int m = (cc * rbs + (uu >> 2)) * 16 + (lane & 15);
        // (the aligned WORD that holds the row's index byte, picked apart where it is used: a byte load is zero-extended
        //  right behind the load - a full vmcnt(0) drain, stores included, in the middle of the pipeline)
        if constexpr (GATHER) return reinterpret_cast<const unsigned*>(x.idx)[m >> 2];
        else return __float_as_uint(x.xs[m]);
    };
    const int ksh = (lane & 3) * 8;
    f32x4 w[GATHER ? 1 : TPS], bs[GATHER ? 1 : TPS];
    mvae_u32x2 d[GATHER ? TPS : 1], dn[GATHER ? TPS : 1];
    int c = 0, u = wave, c1 = 0, u1 = wave + nw;
    if (u1 >= units) { u1 = wave; c1 = 1; }
    unsigned k0 = key_of(c, u), k1 = c1 < nchunks ? key_of(c1, u1) : 0u;
    if constexpr (GATHER) {
#pragma unroll
        for (int j = 0; j < TPS; ++j) d[j] = *reinterpret_cast<const mvae_u32x2*>(table + (size_t)((k0 >> ksh) & 255u) * N + j * 16);
    } else {
#pragma unroll
        for (int j = 0; j < TPS; ++j) {
            w[j] = *reinterpret_cast<const f32x4*>(x.w + tn0 * 16 + col + j * 16);
            bs[j] = *reinterpret_cast<const f32x4*>(x.bias + tn0 * 16 + col + j * 16);
        }
    }
    while (c < nchunks) {
        int c2 = c1, u2 = u1 + nw;
        if (u2 >= units) { u2 = wave; ++c2; }
        unsigned k2 = 0;
        if (c1 < nchunks) {
            if constexpr (GATHER) {
#pragma unroll
                for (int j = 0; j < TPS; ++j) dn[j] = *reinterpret_cast<const mvae_u32x2*>(table + (size_t)((k1 >> ksh) & 255u) * N + j * 16);
            }
            if (c2 < nchunks) k2 = key_of(c2, u2);
        }
        const unsigned char* cbase = reinterpret_cast<const unsigned char*>(x.out) + (size_t)c * chunk_bytes;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(cbase), 0, -1, 0x00020000);
        const int off = ((u >> 2) * ntn + tn0) * 512 + lane * 8;
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < TPS; ++j) __builtin_amdgcn_raw_buffer_store_b64(d[j], rs, off + j * 512, 0, 16 /* sc1: write-through */);
        } else {
            const float xv = __uint_as_float(k0);
#pragma unroll
            for (int j = 0; j < TPS; ++j) store4_bf16_wt(cbase, (unsigned)(off + j * 512), xv * w[j] + bs[j]);
        }
        if (c1 != c) wave_signal_done<false>(x.chunk_done + c);        // (that was this wave's last unit of chunk c)
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < TPS; ++j) d[j] = dn[j];
        }
        k0 = k1; k1 = k2; c = c1; u = u1; c1 = c2; u1 = u2;
    }
}
template <int WPB>      // waves per workgroup of the launch
__device__ __forceinline__ void xpand_body(const mvae_xpand_args x, const int bid, const int nb) {
    const int ntn = x.N >> 4;            // (64 or 48: mvae_rnn_fwd_multi takes nothing else)
    const int lane = (int)(threadIdx.x & 63), wave = bid * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = nb * WPB;
    if (ntn == 64) {            // (G*H = 1024: LSTM, H = 256)
        if (x.idx) xpand_pipe<16, true>(x, wave, nw, lane); else xpand_pipe<16, false>(x, wave, nw, lane);
        return;
    }
    // (768: GRU)
    if (x.idx) xpand_pipe<12, true>(x, wave, nw, lane); else xpand_pipe<12, false>(x, wave, nw, lane);
}
