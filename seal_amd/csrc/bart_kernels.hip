// Fused kernels of the BART step decoder (include/sealnn.h).  Shapes are tiny (one new
// position, <= 16 cached positions, <= 64 encoder positions, head_dim 64): the point is to replace
// ~17 launch-bound PyTorch kernels per decoder layer with 3, not to reach a roofline.
// Every kernel is a template over the STORAGE type (fp32 as the reference runs BART; bf16 for BASELINE.json's
// configs[4]): loads widen to fp32, all arithmetic and every accumulation is fp32, stores round to nearest even.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <type_traits>

#include "../../include/sealnn.h"
#include "fmi_internal.h"

typedef __hip_bfloat16 bf16;
template <typename T> static __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16 *p) { return __bfloat162float(*p); }
template <typename T> static __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16>(bf16 *p, float v) { *p = __float2bfloat16(v); }
// elements 4i .. 4i+3 of a row (16-byte / 8-byte aligned): one vector load
template <typename T> static __device__ __forceinline__ float4 ld4(const T *row, uint32_t i);
template <> __device__ __forceinline__ float4 ld4<float>(const float *row, uint32_t i) { return reinterpret_cast<const float4 *>(row)[i]; }
template <> __device__ __forceinline__ float4 ld4<bf16>(const bf16 *row, uint32_t i)
{
    const uint2 w = reinterpret_cast<const uint2 *>(row)[i];
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
}
template <typename T> static __device__ __forceinline__ void st4(T *row, uint32_t i, float4 v);
template <> __device__ __forceinline__ void st4<float>(float *row, uint32_t i, float4 v) { reinterpret_cast<float4 *>(row)[i] = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16 *row, uint32_t i, float4 v)
{
    bf16 *o = row + 4 * (uint64_t)i;
    o[0] = __float2bfloat16(v.x); o[1] = __float2bfloat16(v.y); o[2] = __float2bfloat16(v.z); o[3] = __float2bfloat16(v.w);
}

// the split-GEMM operand planes of four consecutive elements of a row ([hi | hi | lo * 2^11] in fp16, row length d; see k_split_planes):
// returns whether one of them is finite and beyond fp16's range
// pairs != 0: the row is [hi of 32 columns | lo * 2^11 of the same 32 columns] per 128-byte line, 2 d halves long (k_hgemm_nt's PAIRS operand)
static __device__ __forceinline__ bool store_planes4(__half *prow, uint32_t d, uint32_t i, float4 v, uint32_t pairs = 0)
{
    float in[4] = {v.x, v.y, v.z, v.w};
    unsigned short hi[4], lo[4];
    bool over = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // ONE value of the element and ONE conversion of it feed both planes: the element and the 16 bits of hi go through opaque
        // register moves, so that the stored hi and the hi that lo is the remainder against cannot come from two evaluations (seen on
        // gfx950 inside k_gelu_planes: where gelu(x) sits on a tie between two fp16 values the stored hi was the even neighbour, the
        // subtracted one the other -- lo came out with the wrong sign on 8 of 315 392 elements)
        asm volatile("" : "+v"(in[j]));
        const float a = fabsf(in[j]);
        over |= a > 65504.f && a < __builtin_huge_valf();
        unsigned short hb = __half_as_ushort(__float2half_rn(in[j]));
        asm volatile("" : "+v"(hb));
        hi[j] = hb;
        lo[j] = __half_as_ushort(__float2half_rn((in[j] - __half2float(__ushort_as_half(hb))) * 2048.f));
    }
    __half *o = prow + 4 * (uint64_t)i;
    const uint2 hw = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
    const uint2 lw = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    if (pairs) {
        __half *q = prow + ((4 * i) >> 5) * 64 + ((4 * i) & 31u);
        *(uint2 *)q = hw;
        *(uint2 *)(q + 32) = lw;
        return over;
    }
    *(uint2 *)o = hw;
    *(uint2 *)(o + d) = hw;
    *(uint2 *)(o + 2 * (uint64_t)d) = lw;
    return over;
}

// the same for ONE element (lane = column: the wave's three stores are 128 contiguous bytes each)
static __device__ __forceinline__ bool store_planes1(__half *prow, uint32_t d, uint32_t col, float v, uint32_t pairs = 0)
{
    asm volatile("" : "+v"(v));
    const float a = fabsf(v);
    const bool over = a > 65504.f && a < __builtin_huge_valf();
    unsigned short hb = __half_as_ushort(__float2half_rn(v));
    asm volatile("" : "+v"(hb));
    const unsigned short lb = __half_as_ushort(__float2half_rn((v - __half2float(__ushort_as_half(hb))) * 2048.f));
    unsigned short *o = reinterpret_cast<unsigned short *>(prow);
    if (pairs) { o[(col >> 5) * 64 + (col & 31u)] = hb; o[(col >> 5) * 64 + 32 + (col & 31u)] = lb; return over; }
    o[col] = hb; o[(uint64_t)d + col] = hb; o[2 * (uint64_t)d + col] = lb;
    return over;
}

static __device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// the value lane `i` holds, for a wave-uniform i: v_readlane into a scalar register (no LDS round trip, unlike __shfl,
// which is a ds_bpermute)
static __device__ __forceinline__ float lane_value(float v, uint32_t i)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)i));
}
static __device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// one wavefront per (row, head); lane = head dimension.  `anc` (optional, [T][rows]): the cache is
// never reordered when beams are re-ranked; instead anc[p][row] names the row whose slot at position p
// belongs to `row`'s history (the decoder permutes this 20 KB table, not the 100+ MB cache).
// `pb` (optional, [3 * heads * 64]) / `pa`: qkv holds the raw accumulators of a split GEMM (seal_amd/split_gemm.py: Deferred) and the
// projection's output is pa * qkv + pb -- the GEMM's own epilogue, applied here on read instead of in a pass over qkv
template <typename T_>
__global__ __launch_bounds__(256) void k_self_attn_step(const T_ *qkv, T_ *kcache, T_ *vcache, const int64_t *d_t,
                                                        uint32_t rows, uint32_t heads, uint32_t T, float scale, T_ *out,
                                                        int32_t *anc, const T_ *pb, float pa, uint32_t n_slabs = 1, uint64_t slab_stride = 0,
                                                        __half *oplanes = nullptr, uint32_t *flag = nullptr, uint32_t pairs = 0)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rows * heads) return;
    const uint32_t row = item / heads, head = item % heads;
    const uint32_t t = (uint32_t)*d_t;
    const T_ *base = qkv + ((uint64_t)row * 3 * heads + head) * 64;
    float q = ldf(base + lane);
    float kn = ldf(base + (uint64_t)heads * 64 + lane);       // (bf16: already the value the cache will hold)
    float vn = ldf(base + (uint64_t)2 * heads * 64 + lane);
    // (n_slabs > 1: qkv holds the slabs of a split-K product, sealnn_hgemm_nt: added here, in slab order)
    // (four slabs' loads are in flight together -- a loop of load, add, load, ... is one memory round trip per slab --; a slab beyond the last
    //  reads slab 0 again and is not added)
    for (uint32_t sl = 1; sl < n_slabs; sl += 4) {
        float eq[4], ek[4], ev[4];
#pragma unroll
        for (uint32_t s = 0; s < 4; s++) {
            const T_ *bs = base + (uint64_t)(sl + s < n_slabs ? sl + s : 0) * slab_stride;
            eq[s] = ldf(bs + lane); ek[s] = ldf(bs + (uint64_t)heads * 64 + lane); ev[s] = ldf(bs + (uint64_t)2 * heads * 64 + lane);
        }
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (sl + s < n_slabs) { q += eq[s]; kn += ek[s]; vn += ev[s]; }
    }
    if (pb) {
        const T_ *b = pb + head * 64 + lane;
        q = pa * q + ldf(b); kn = pa * kn + ldf(b + (uint64_t)heads * 64); vn = pa * vn + ldf(b + (uint64_t)2 * heads * 64);
    }
    q *= scale;
    T_ *kc = kcache + ((uint64_t)row * heads + head) * T * 64;
    T_ *vc = vcache + ((uint64_t)row * heads + head) * T * 64;
    stf(kc + (uint64_t)t * 64 + lane, kn);
    stf(vc + (uint64_t)t * 64 + lane, vn);
    if (anc && head == 0 && lane == 0) anc[(uint64_t)t * rows + row] = (int32_t)row;
    // The history of the row: position p sits in the slot of row anc[p][row].  All of it is fetched BEFORE anything is
    // computed -- lane p reads the ancestor of position p, then every K / V load of the wave is issued back to back (their
    // addresses no longer hang on a load inside the loop: the serial chain of <= 16 dependent round trips was the kernel's time)
    // (no branch around the loads: with `if (p < t)` the compiler waited for every position's pair before issuing the next one, up to
    //  sixteen round trips in a row.  A position beyond the history re-reads the last one -- a line the wave has just asked for -- and is
    //  never used below.)
    const uint32_t my_anc = (anc && lane < t) ? (uint32_t)anc[(uint64_t)lane * rows + row] : row;
    const uint32_t last = t ? t - 1 : 0;
    float kreg[FMI_MAX_LEVELS], vreg[FMI_MAX_LEVELS];
#pragma unroll
    for (uint32_t p = 0; p < FMI_MAX_LEVELS; p++) {
        const uint32_t pc = p < t ? p : last;                         // wave-uniform
        const uint32_t a_p = (uint32_t)__builtin_amdgcn_readlane((int)my_anc, (int)pc);
        const uint64_t src = ((uint64_t)a_p * heads + head) * T * 64 + (uint64_t)pc * 64 + lane;
        kreg[p] = ldf(kcache + src);
        vreg[p] = ldf(vcache + src);
    }
    // scores over positions 0..t (the new one from registers), in position order as before
    float s[FMI_MAX_LEVELS];          // T <= 17 positions kept in registers
    float m = -__builtin_huge_valf();
#pragma unroll
    for (uint32_t p = 0; p < FMI_MAX_LEVELS; p++) {
        s[p] = 0.f;
        if (p <= t) {
            const float d = wave_sum(q * (p == t ? kn : kreg[p]));
            s[p] = d;
            m = fmaxf(m, d);
        }
    }
    float denom = 0.f, acc = 0.f;
#pragma unroll
    for (uint32_t p = 0; p < FMI_MAX_LEVELS; p++) {
        if (p <= t) {
            const float e = expf(s[p] - m);
            denom += e;
            acc += e * (p == t ? vn : vreg[p]);
        }
    }
    const float o = acc / denom;
    if (out) stf(out + (uint64_t)row * heads * 64 + head * 64 + lane, o);
    // (oplanes: the output as the split operand of the projection that follows -- [hi | hi | lo * 2^11] in fp16 --, written here instead of by a
    //  pass of k_split_planes over `out`)
    if (oplanes && __any((int)store_planes1(oplanes + (uint64_t)row * (pairs ? 2 : 3) * heads * 64, heads * 64, head * 64 + lane, o, pairs)) && lane == 0 && flag) atomicAdd(flag, 1u);
}

// K [64, S] and V [S, 64] of one (query, head) into LDS (s_k, s_v: 16-byte aligned, n = 64 * S elements each).  Every load of a thread is
// issued before its first store: as a plain copy loop the compiler waited for each pair of loads before storing it, eight memory round
// trips in a row at the head of a 17 us kernel.
template <typename T_>
static __device__ __forceinline__ void stage_kv(const T_ *k, const T_ *v, float *s_k, float *s_v, uint32_t n)
{
    if constexpr (std::is_same<T_, float>::value) {
        const uint32_t n4 = n / 4;
        const float4 *k4 = reinterpret_cast<const float4 *>(k), *v4 = reinterpret_cast<const float4 *>(v);
        float4 *sk4 = reinterpret_cast<float4 *>(s_k), *sv4 = reinterpret_cast<float4 *>(s_v);
        for (uint32_t i0 = threadIdx.x; i0 < n4; i0 += 4 * blockDim.x) {
            float4 a[4], b[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t i = i0 + j * blockDim.x;
                a[j] = b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < n4) { a[j] = k4[i]; b[j] = v4[i]; }
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t i = i0 + j * blockDim.x;
                if (i < n4) { sk4[i] = a[j]; sv4[i] = b[j]; }
            }
        }
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { s_k[i] = ldf(k + i); s_v[i] = ldf(v + i); }
    }
}

// one row's attention over the S encoder positions of its (query, head): scores with lane = encoder position, output with lane = dim.
// K / V through any pointer (LDS or global).  The lanes beyond S read position 0 and are masked afterwards: no branch inside the loops,
// so the reads of a loop are issued together (with `if (lane < S)` around each, every one of the 128 reads was waited for on its own).
template <typename KV>
static __device__ __forceinline__ float cross_attn_row(float qd, const KV *k, const KV *v, float bi, uint32_t S, uint32_t lane)
{
    const uint32_t l = lane < S ? lane : 0;
    float sc = 0.f;
#pragma unroll 16
    for (uint32_t d = 0; d < 64; d++) sc += lane_value(qd, d) * ldf(k + d * S + l);
    sc = lane < S ? sc + bi : -__builtin_huge_valf();
    const float m = wave_max(sc);
    const float e = lane < S ? expf(sc - m) : 0.f;
    const float denom = wave_sum(e);
    float acc = 0.f;
#pragma unroll 8
    for (uint32_t p = 0; p < S; p++) acc += lane_value(e, p) * ldf(v + p * 64 + lane);
    return acc / denom;
}

// one workgroup per (query, head): the encoder K [64, S] and V [S, 64] of that head are staged in LDS once
// and shared by the query's beams (one wavefront per beam; scores: lane = encoder position, output: lane = dim)
template <typename T_>
__global__ __launch_bounds__(1024) void k_cross_attn_step(const T_ *q, const T_ *ck, const T_ *cv, const T_ *bias,
                                                          uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S, float scale,
                                                          T_ *out, const T_ *qb = nullptr, float qa = 1.f, uint32_t n_slabs = 1, uint64_t slab_stride = 0,
                                                          __half *oplanes = nullptr, uint32_t *flag = nullptr, uint32_t pairs = 0)
{
    __shared__ __attribute__((aligned(16))) float s_k[64 * 64];
    __shared__ __attribute__((aligned(16))) float s_v[64 * 64];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t b = blockIdx.x / heads, head = blockIdx.x % heads;
    const T_ *k = ck + ((uint64_t)b * heads + head) * 64 * S;      // [64, S]
    const T_ *v = cv + ((uint64_t)b * heads + head) * S * 64;      // [S, 64]
    stage_kv(k, v, s_k, s_v, 64 * S);
    const float bi = lane < S ? ldf(bias + (uint64_t)b * S + lane) : 0.f;
    __syncthreads();
    for (uint32_t beam = wv; beam < beams; beam += nw) {
        const uint32_t row = b * beams + beam;
        // (qb / qa / slabs: q = the raw accumulators of the query projection, possibly as split-K slabs: the projection is qa * sum + qb)
        float qv = ldf(q + ((uint64_t)row * heads + head) * 64 + lane);
        for (uint32_t sl = 1; sl < n_slabs; sl += 4) {       // (four slabs' loads in flight together: k_self_attn_step)
            float e[4];
#pragma unroll
            for (uint32_t s = 0; s < 4; s++) e[s] = ldf(q + (uint64_t)(sl + s < n_slabs ? sl + s : 0) * slab_stride + ((uint64_t)row * heads + head) * 64 + lane);
#pragma unroll
            for (uint32_t s = 0; s < 4; s++)
                if (sl + s < n_slabs) qv += e[s];
        }
        if (qb) qv = qa * qv + ldf(qb + head * 64 + lane);
        const float o = cross_attn_row(qv * scale, s_k, s_v, bi, S, lane);
        if (out) stf(out + (uint64_t)row * heads * 64 + head * 64 + lane, o);
        if (oplanes && __any((int)store_planes1(oplanes + (uint64_t)row * (pairs ? 2 : 3) * heads * 64, heads * 64, head * 64 + lane, o, pairs)) && lane == 0 && flag) atomicAdd(flag, 1u);
    }
}

// teacher-forced causal self-attention over T positions: one wavefront per (sequence, head); lane = dim
template <typename T_>
__global__ __launch_bounds__(256) void k_causal_self_attn(const T_ *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale,
                                                          T_ *out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_seq * heads) return;
    const uint32_t n = item / heads, head = item % heads;
    const uint64_t stride = (uint64_t)3 * heads * 64;               // per position
    const T_ *base = qkv + (uint64_t)n * T * stride + head * 64;
    float kreg[FMI_MAX_LEVELS], vreg[FMI_MAX_LEVELS];
    for (uint32_t j = 0; j < T; j++) {
        kreg[j] = ldf(base + (uint64_t)j * stride + (uint64_t)heads * 64 + lane);
        vreg[j] = ldf(base + (uint64_t)j * stride + (uint64_t)2 * heads * 64 + lane);
    }
    for (uint32_t i = 0; i < T; i++) {
        const float q = ldf(base + (uint64_t)i * stride + lane) * scale;
        float s[FMI_MAX_LEVELS];
        float m = -__builtin_huge_valf();
        for (uint32_t j = 0; j <= i; j++) { s[j] = wave_sum(q * kreg[j]); m = fmaxf(m, s[j]); }
        float denom = 0.f, acc = 0.f;
        for (uint32_t j = 0; j <= i; j++) { const float e = expf(s[j] - m); denom += e; acc += e * vreg[j]; }
        stf(out + ((uint64_t)n * T + i) * heads * 64 + head * 64 + lane, acc / denom);
    }
}

// teacher-forced self-attention over a prefix TREE: every node (one decoder position of one distinct prefix) attends
// its ancestors and itself, anc[node][0 .. depth]: node indices from the root down, -1 beyond the node's depth.  One
// wavefront per (node, head); lane = dim.  Same arithmetic in the same order as k_causal_self_attn does for position
// `depth` of a row holding that prefix.
template <typename T_>
__global__ __launch_bounds__(256) void k_tree_self_attn(const T_ *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t A, uint32_t heads,
                                                        float scale, T_ *out, const T_ *pb, float pa)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_nodes * heads) return;
    const uint32_t node = item / heads, head = item % heads;
    const uint64_t stride = (uint64_t)3 * heads * 64;               // per node
    // (pb / pa: qkv = raw split-GEMM accumulators, the projection is pa * qkv + pb; see k_self_attn_step)
    float q = ldf(qkv + (uint64_t)node * stride + head * 64 + lane);
    float bk = 0.f, bv = 0.f;
    if (pb) {
        const T_ *b = pb + head * 64 + lane;
        q = pa * q + ldf(b); bk = ldf(b + (uint64_t)heads * 64); bv = ldf(b + (uint64_t)2 * heads * 64);
    }
    q *= scale;
    const int32_t *mine = anc + (uint64_t)node * A;
    float s[FMI_MAX_LEVELS], vreg[FMI_MAX_LEVELS];
    float m = -__builtin_huge_valf();
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t j = 0; j < FMI_MAX_LEVELS; j++) {
        s[j] = 0.f; vreg[j] = 0.f;
        const int32_t a = j < A ? mine[j] : -1;
        if (a >= 0 && cnt == j) {
            const T_ *row = qkv + (uint64_t)a * stride + head * 64 + lane;
            float k = ldf(row + (uint64_t)heads * 64);
            vreg[j] = ldf(row + (uint64_t)2 * heads * 64);
            if (pb) { k = pa * k + bk; vreg[j] = pa * vreg[j] + bv; }
            s[j] = wave_sum(q * k);
            m = fmaxf(m, s[j]);
            cnt = j + 1;
        }
    }
    float denom = 0.f, acc = 0.f;
#pragma unroll
    for (uint32_t j = 0; j < FMI_MAX_LEVELS; j++)
        if (j < cnt) { const float e = expf(s[j] - m); denom += e; acc += e * vreg[j]; }
    stf(out + (uint64_t)node * heads * 64 + head * 64 + lane, acc / denom);
}

// cross-attention for arbitrary rows: row_batch[row] selects the query whose encoder K/V to use
template <typename T_>
__global__ __launch_bounds__(256) void k_cross_attn_rows(const T_ *q, const T_ *ck, const T_ *cv, const T_ *bias,
                                                         const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S,
                                                         float scale, T_ *out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rows * heads) return;
    const uint32_t row = item / heads, head = item % heads, b = (uint32_t)row_batch[row];
    const float qd = ldf(q + ((uint64_t)row * heads + head) * 64 + lane) * scale;
    const T_ *k = ck + ((uint64_t)b * heads + head) * 64 * S;
    const T_ *v = cv + ((uint64_t)b * heads + head) * S * 64;
    const float bi = lane < S ? ldf(bias + (uint64_t)b * S + lane) : 0.f;
    stf(out + (uint64_t)row * heads * 64 + head * 64 + lane, cross_attn_row(qd, k, v, bi, S, lane));
}

// the same for rows that come in runs: one workgroup per (`group` consecutive rows, head) stages the K [64, S] and V [S, 64] of the FIRST
// row's query in LDS once and its waves walk the rows -- a row of that query (teacher forcing: the T positions of a sequence; the nodes of a
// query's prefix tree, which are consecutive: all of them but at the borders between queries) reads LDS, a row of another query reads
// its own K / V from global memory as k_cross_attn_rows does: 16 x fewer bytes out of L2 for the 3 200-row rescoring forward, whose
// one-wave-per-(row, head) form spent 84 us per launch re-reading 32 KB per wave.  Same arithmetic, in the same order, on every path.
template <typename T_>
__global__ __launch_bounds__(512) void k_cross_attn_runs(const T_ *q, const T_ *ck, const T_ *cv, const T_ *bias,
                                                         const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads,
                                                         uint32_t S, float scale, T_ *out)
{
    extern __shared__ float4 s_kv4[];               // K [64, S] then V [S, 64]: 512 S bytes, so that several runs share a CU
    float *s_k = reinterpret_cast<float *>(s_kv4), *s_v = s_k + 64 * S;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t run = blockIdx.x / heads, head = blockIdx.x % heads;
    const uint32_t row0 = run * group;
    const uint32_t b = (uint32_t)row_batch[row0];
    stage_kv(ck + ((uint64_t)b * heads + head) * 64 * S, cv + ((uint64_t)b * heads + head) * S * 64, s_k, s_v, 64 * S);
    const float bi = lane < S ? ldf(bias + (uint64_t)b * S + lane) : 0.f;
    __syncthreads();
    for (uint32_t t = wv; t < group && row0 + t < rows; t += nw) {
        const uint32_t row = row0 + t;
        const uint32_t rb = (uint32_t)row_batch[row];
        const float qd = ldf(q + ((uint64_t)row * heads + head) * 64 + lane) * scale;
        float o;
        if (rb == b) {
            o = cross_attn_row(qd, s_k, s_v, bi, S, lane);
        } else {
            const float bo = lane < S ? ldf(bias + (uint64_t)rb * S + lane) : 0.f;
            o = cross_attn_row(qd, ck + ((uint64_t)rb * heads + head) * 64 * S, cv + ((uint64_t)rb * heads + head) * S * 64, bo, S, lane);
        }
        stf(out + (uint64_t)row * heads * 64 + head * 64 + lane, o);
    }
}

// one wavefront per row, d <= 4096 (16 float4 per lane)
template <typename T_>
// (yb / ya: y = raw split-GEMM accumulators, the addend is ya * y + yb; see k_self_attn_step)
__global__ __launch_bounds__(256) void k_add_layernorm(const T_ *x, const T_ *y, const T_ *gamma, const T_ *beta,
                                                       uint32_t rows, uint32_t d, float eps, T_ *out, __half *planes, uint32_t *flag,
                                                       const T_ *yb, float ya, uint32_t y_slabs, uint64_t y_slab_stride, uint32_t pairs)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T_ *xr = x + (uint64_t)row * d, *yr = y + (uint64_t)row * d;
    const uint32_t n4 = d / 4;
    // The row in chunks of four float4 per lane (d = 1024: one chunk).  The loads of a chunk are issued together, unconditionally -- a lane
    // beyond the row reads element 0 and its value is dropped: with `if (i < n4)` around each, every pair of loads was waited for before the
    // next was issued, one memory round trip per 256 columns -- and gamma / beta of the first chunk are asked for BEFORE the reductions.
    const T_ *ybp = yb ? yb : gamma;                  // (a valid address when there is no epilogue to apply: loaded, not used)
    float4 v[16];                                    // fully unrolled below: registers, not scratch
    float4 g0[4], b0[4];
    float sum = 0.f;
#pragma unroll
    for (uint32_t j0 = 0; j0 < 16; j0 += 4) {
        if (64 * j0 >= n4) {                         // (wave-uniform) chunks beyond the row
#pragma unroll
            for (uint32_t jj = 0; jj < 4; jj++) v[j0 + jj] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        float4 a[4], b[4], c[4];
#pragma unroll
        for (uint32_t jj = 0; jj < 4; jj++) {
            const uint32_t i = lane + 64 * (j0 + jj), ic = i < n4 ? i : 0;
            a[jj] = ld4(xr, ic); b[jj] = ld4(yr, ic); c[jj] = ld4(ybp, ic);
            if (j0 == 0) { g0[jj] = ld4(gamma, ic); b0[jj] = ld4(beta, ic); }
        }
        // (y_slabs > 1: y holds the slabs of a split-K product, sealnn_hgemm_nt -- summed here, in slab order, as they are read)
        // (four slabs' loads are in flight together; a slab beyond the last reads slab 0 again and is not added)
        for (uint32_t sl = 1; sl < y_slabs; sl += 4) {
            float4 e[4][4];
#pragma unroll
            for (uint32_t s = 0; s < 4; s++) {
                const T_ *ys = yr + (uint64_t)(sl + s < y_slabs ? sl + s : 0) * y_slab_stride;
#pragma unroll
                for (uint32_t jj = 0; jj < 4; jj++) {
                    const uint32_t i = lane + 64 * (j0 + jj), ic = i < n4 ? i : 0;
                    e[s][jj] = ld4(ys, ic);
                }
            }
#pragma unroll
            for (uint32_t s = 0; s < 4; s++) {
                if (sl + s >= y_slabs) continue;
#pragma unroll
                for (uint32_t jj = 0; jj < 4; jj++)
                    b[jj] = make_float4(b[jj].x + e[s][jj].x, b[jj].y + e[s][jj].y, b[jj].z + e[s][jj].z, b[jj].w + e[s][jj].w);
            }
        }
#pragma unroll
        for (uint32_t jj = 0; jj < 4; jj++) {
            const uint32_t i = lane + 64 * (j0 + jj);
            float4 bb = b[jj];
            if (yb) bb = make_float4(ya * bb.x + c[jj].x, ya * bb.y + c[jj].y, ya * bb.z + c[jj].z, ya * bb.w + c[jj].w);
            const float4 w = make_float4(a[jj].x + bb.x, a[jj].y + bb.y, a[jj].z + bb.z, a[jj].w + bb.w);
            const bool in = i < n4;
            v[j0 + jj] = in ? w : make_float4(0.f, 0.f, 0.f, 0.f);
            sum += in ? (w.x + w.y) + (w.z + w.w) : 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float var = 0.f;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        if (lane + 64 * j < n4) {
            const float a = v[j].x - mean, b = v[j].y - mean, e = v[j].z - mean, f = v[j].w - mean;
            var += (a * a + b * b) + (e * e + f * f);
        }
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)d + eps);
    T_ *orow = out + (uint64_t)row * d;
    __half *prow = planes ? planes + (uint64_t)row * (pairs ? 2 : 3) * d : nullptr;      // (fp32 only) the same values as the next GEMM's split operand
    bool over = false;
#pragma unroll
    for (uint32_t j0 = 0; j0 < 16; j0 += 4) {
        if (64 * j0 >= n4) continue;
        float4 g[4], bb[4];
#pragma unroll
        for (uint32_t jj = 0; jj < 4; jj++) {
            const uint32_t i = lane + 64 * (j0 + jj), ic = i < n4 ? i : 0;
            if (j0 == 0) { g[jj] = g0[jj]; bb[jj] = b0[jj]; }
            else { g[jj] = ld4(gamma, ic); bb[jj] = ld4(beta, ic); }
        }
#pragma unroll
        for (uint32_t jj = 0; jj < 4; jj++) {
            const uint32_t j = j0 + jj, i = lane + 64 * j;
            if (i < n4) {
                const float4 r = make_float4((v[j].x - mean) * rstd * g[jj].x + bb[jj].x, (v[j].y - mean) * rstd * g[jj].y + bb[jj].y,
                                             (v[j].z - mean) * rstd * g[jj].z + bb[jj].z, (v[j].w - mean) * rstd * g[jj].w + bb[jj].w);
                st4(orow, i, r);
                if (prow) over |= store_planes4(prow, d, i, r, pairs);
            }
        }
    }
    if (over && flag) atomicAdd(flag, 1u);
}

// fc2's operand: planes of gelu(x) (the erf form, the arithmetic of torch's GeluCUDAKernelImpl: 0.5 * x * (1 + erf(x * M_SQRT1_2)))
// (xb / xa: x = raw split-GEMM accumulators, gelu's argument is xa * x + xb; see k_self_attn_step)
__global__ __launch_bounds__(256) void k_gelu_planes(const float *x, uint32_t rows, uint32_t d, __half *planes, uint32_t *flag, const float *xb, float xa,
                                                     uint32_t n_slabs, uint64_t slab_stride, uint32_t pairs)
{
    const uint32_t per_row = d / 4;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rows * per_row) return;
    const uint32_t r = (uint32_t)(i / per_row), c = (uint32_t)(i % per_row);
    float4 v = reinterpret_cast<const float4 *>(x + (uint64_t)r * d)[c];
    // (n_slabs > 1: x holds the slabs of a split-K product, sealnn_hgemm_nt -- added here, in slab order)
    for (uint32_t sl = 1; sl < n_slabs; sl += 4) {           // (four slabs' loads in flight together: k_self_attn_step)
        float4 e[4];
#pragma unroll
        for (uint32_t s = 0; s < 4; s++) e[s] = reinterpret_cast<const float4 *>(x + (uint64_t)(sl + s < n_slabs ? sl + s : 0) * slab_stride + (uint64_t)r * d)[c];
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (sl + s < n_slabs) v = make_float4(v.x + e[s].x, v.y + e[s].y, v.z + e[s].z, v.w + e[s].w);
    }
    if (xb) { const float4 b = reinterpret_cast<const float4 *>(xb)[c]; v = make_float4(xa * v.x + b.x, xa * v.y + b.y, xa * v.z + b.z, xa * v.w + b.w); }
    const float kAlpha = 0.70710678118654752440f;
    const float4 g = make_float4(0.5f * v.x * (1.f + erff(v.x * kAlpha)), 0.5f * v.y * (1.f + erff(v.y * kAlpha)),
                                 0.5f * v.z * (1.f + erff(v.z * kAlpha)), 0.5f * v.w * (1.f + erff(v.w * kAlpha)));
    if (store_planes4(planes + (uint64_t)r * (pairs ? 2 : 3) * d, d, c, g, pairs) && flag) atomicAdd(flag, 1u);
}

// alpha * (slab 0 + slab 1 + ...) + bias -> a finished fp32 product: for the consumers that are not kernels of this file (torch's fused attention in
// the encoder, the permutes of the cross-attention K / V), so that the hand-written product serves them too
__global__ __launch_bounds__(256) void k_finish_product(const float *acc, uint32_t n_slabs, uint64_t slab_stride, const float *bias, float alpha, uint64_t n4,
                                                        uint32_t per_row, float *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4 *>(acc)[i];
    for (uint32_t sl = 1; sl < n_slabs; sl += 4) {           // (four slabs' loads in flight together: k_self_attn_step)
        float4 e[4];
#pragma unroll
        for (uint32_t s = 0; s < 4; s++) e[s] = reinterpret_cast<const float4 *>(acc + (uint64_t)(sl + s < n_slabs ? sl + s : 0) * slab_stride)[i];
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (sl + s < n_slabs) v = make_float4(v.x + e[s].x, v.y + e[s].y, v.z + e[s].z, v.w + e[s].w);
    }
    const float4 b = reinterpret_cast<const float4 *>(bias)[i % per_row];
    reinterpret_cast<float4 *>(out)[i] = make_float4(alpha * v.x + b.x, alpha * v.y + b.y, alpha * v.z + b.z, alpha * v.w + b.w);
}

#define NNCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { fmi_set_error("sealnn launch failed: %s", hipGetErrorString(e_)); return FMI_ERR_HIP; } } while (0)

template <typename T_>
static int self_attn_step(void *stream, const void *qkv, void *kcache, void *vcache, const int64_t *d_t, uint32_t rows,
                          uint32_t heads, uint32_t T, float scale, void *out, int32_t *anc, const void *pb = nullptr, float pa = 1.f)
{
    if (T > FMI_MAX_LEVELS) { fmi_set_error("sealnn_self_attn_step: at most %u cached positions", FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    const uint32_t items = rows * heads;
    hipLaunchKernelGGL(k_self_attn_step<T_>, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T_ *)qkv, (T_ *)kcache, (T_ *)vcache,
                       d_t, rows, heads, T, scale, (T_ *)out, anc, (const T_ *)pb, pa);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int cross_attn_step(void *stream, const void *q, const void *ck, const void *cv, const void *bias, uint32_t batch,
                           uint32_t beams, uint32_t heads, uint32_t S, float scale, void *out)
{
    if (S > 64) { fmi_set_error("sealnn_cross_attn_step: encoder length %u > 64", S); return FMI_ERR_UNSUPPORTED; }
    // eight waves per workgroup (a wave takes every eighth beam): 512 threads and 32 KB of LDS let four workgroups share a CU, so the
    // 640 workgroups of a 40-query step are resident at once (fifteen waves: two per CU, two rounds)
    const uint32_t waves = beams < 8 ? beams : 8;
    hipLaunchKernelGGL(k_cross_attn_step<T_>, dim3(batch * heads), dim3(waves * 64), 0, (hipStream_t)stream, (const T_ *)q, (const T_ *)ck,
                       (const T_ *)cv, (const T_ *)bias, batch, beams, heads, S, scale, (T_ *)out);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int add_layernorm(void *stream, const void *x, const void *y, const void *gamma, const void *beta, uint32_t rows,
                         uint32_t d, float eps, void *out, void *planes = nullptr, uint32_t *flag = nullptr, const void *yb = nullptr, float ya = 1.f,
                         uint32_t y_slabs = 1, uint64_t y_slab_stride = 0, uint32_t pairs = 0)
{
    if (d % 4 || d > 4096 || (pairs && d % 32)) { fmi_set_error("sealnn_add_layernorm: d=%u unsupported", d); return FMI_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_add_layernorm<T_>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T_ *)x, (const T_ *)y,
                       (const T_ *)gamma, (const T_ *)beta, rows, d, eps, (T_ *)out, (__half *)planes, flag, (const T_ *)yb, ya, y_slabs, y_slab_stride, pairs);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int causal_self_attn(void *stream, const void *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale, void *out)
{
    if (T > FMI_MAX_LEVELS) { fmi_set_error("sealnn_causal_self_attn: at most %u positions", FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    const uint32_t items = n_seq * heads;
    hipLaunchKernelGGL(k_causal_self_attn<T_>, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T_ *)qkv, n_seq, T, heads, scale, (T_ *)out);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int tree_self_attn(void *stream, const void *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t max_depth1, uint32_t heads,
                          float scale, void *out, const void *pb = nullptr, float pa = 1.f)
{
    if (max_depth1 > FMI_MAX_LEVELS) { fmi_set_error("sealnn_tree_self_attn: at most %u positions", FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    const uint32_t items = n_nodes * heads;
    if (!items) return FMI_OK;
    hipLaunchKernelGGL(k_tree_self_attn<T_>, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T_ *)qkv, anc, n_nodes, max_depth1, heads,
                       scale, (T_ *)out, (const T_ *)pb, pa);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int cross_attn_rows(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                           const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S, float scale, void *out)
{
    if (S > 64) { fmi_set_error("sealnn_cross_attn_rows: encoder length %u > 64", S); return FMI_ERR_UNSUPPORTED; }
    const uint32_t items = rows * heads;
    hipLaunchKernelGGL(k_cross_attn_rows<T_>, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const T_ *)q, (const T_ *)ck, (const T_ *)cv,
                       (const T_ *)bias, row_batch, rows, heads, S, scale, (T_ *)out);
    NNCHK();
    return FMI_OK;
}

template <typename T_>
static int cross_attn_runs(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                           const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads, uint32_t S, float scale, void *out)
{
    if (S > 64) { fmi_set_error("sealnn_cross_attn_runs: encoder length %u > 64", S); return FMI_ERR_UNSUPPORTED; }
    if (group == 0) { fmi_set_error("sealnn_cross_attn_runs: runs of 0 rows"); return FMI_ERR_ARG; }
    if (!rows) return FMI_OK;
    // one wave per position of the run, up to eight; the last run may be short
    const unsigned threads = 64 * std::min<unsigned>(8, std::max<unsigned>(1, group));
    hipLaunchKernelGGL(k_cross_attn_runs<T_>, dim3(((rows + group - 1) / group) * heads), dim3(threads), (size_t)512 * S, (hipStream_t)stream, (const T_ *)q,
                       (const T_ *)ck, (const T_ *)cv, (const T_ *)bias, row_batch, rows, group, heads, S, scale, (T_ *)out);
    NNCHK();
    return FMI_OK;
}

// ---- C ABI: fp32 (the reference's arithmetic) and bf16 storage (suffix _bf16; same shapes, 2-byte elements) ----
extern "C" int sealnn_self_attn_step(void *stream, const float *qkv, float *kcache, float *vcache, const int64_t *d_t, uint32_t rows,
                                     uint32_t heads, uint32_t T, float scale, float *out, int32_t *anc)
{ return self_attn_step<float>(stream, qkv, kcache, vcache, d_t, rows, heads, T, scale, out, anc); }
extern "C" int sealnn_self_attn_step_bf16(void *stream, const void *qkv, void *kcache, void *vcache, const int64_t *d_t, uint32_t rows,
                                          uint32_t heads, uint32_t T, float scale, void *out, int32_t *anc)
{ return self_attn_step<bf16>(stream, qkv, kcache, vcache, d_t, rows, heads, T, scale, out, anc); }

extern "C" int sealnn_cross_attn_step(void *stream, const float *q, const float *ck, const float *cv, const float *bias, uint32_t batch,
                                      uint32_t beams, uint32_t heads, uint32_t S, float scale, float *out)
{ return cross_attn_step<float>(stream, q, ck, cv, bias, batch, beams, heads, S, scale, out); }
extern "C" int sealnn_cross_attn_step_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias, uint32_t batch,
                                           uint32_t beams, uint32_t heads, uint32_t S, float scale, void *out)
{ return cross_attn_step<bf16>(stream, q, ck, cv, bias, batch, beams, heads, S, scale, out); }

// the two step kernels between hand-written products: operands as raw (split-K) accumulators with their epilogue, results as split planes
static int self_attn_step_x(void *stream, const float *qkv_acc, uint32_t n_slabs, uint64_t slab_stride, const float *qkv_bias, float alpha,
                                       float *kcache, float *vcache, const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out,
                                       void *out_planes, uint32_t *d_flag, int32_t *anc, uint32_t pairs)
{
    if (T > FMI_MAX_LEVELS) { fmi_set_error("sealnn_self_attn_step_x: at most %u positions", FMI_MAX_LEVELS); return FMI_ERR_UNSUPPORTED; }
    if (!qkv_bias || (!out && !out_planes) || n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_self_attn_step_x: bad argument"); return FMI_ERR_ARG; }
    const uint32_t items = rows * heads;
    hipLaunchKernelGGL(k_self_attn_step<float>, dim3((items + 3) / 4), dim3(256), 0, (hipStream_t)stream, qkv_acc, kcache, vcache, d_t, rows, heads, T, scale,
                       out, anc, qkv_bias, alpha, n_slabs, slab_stride, (__half *)out_planes, d_flag, pairs);
    NNCHK();
    return FMI_OK;
}

extern "C" int sealnn_self_attn_step_x(void *stream, const float *qkv_acc, uint32_t n_slabs, uint64_t slab_stride, const float *qkv_bias, float alpha,
                                       float *kcache, float *vcache, const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out,
                                       void *out_planes, uint32_t *d_flag, int32_t *anc)
{ return self_attn_step_x(stream, qkv_acc, n_slabs, slab_stride, qkv_bias, alpha, kcache, vcache, d_t, rows, heads, T, scale, out, out_planes, d_flag, anc, 0); }
extern "C" int sealnn_self_attn_step_x_pairs(void *stream, const float *qkv_acc, uint32_t n_slabs, uint64_t slab_stride, const float *qkv_bias, float alpha,
                                             float *kcache, float *vcache, const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out,
                                             void *out_planes, uint32_t *d_flag, int32_t *anc)
{ return self_attn_step_x(stream, qkv_acc, n_slabs, slab_stride, qkv_bias, alpha, kcache, vcache, d_t, rows, heads, T, scale, out, out_planes, d_flag, anc, 1); }

static int cross_attn_step_x(void *stream, const float *q_acc, uint32_t n_slabs, uint64_t slab_stride, const float *q_bias, float alpha,
                                        const float *ck, const float *cv, const float *bias, uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S,
                                        float scale, float *out, void *out_planes, uint32_t *d_flag, uint32_t pairs)
{
    if (S > 64) { fmi_set_error("sealnn_cross_attn_step_x: encoder length %u > 64", S); return FMI_ERR_UNSUPPORTED; }
    if ((!out && !out_planes) || n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_cross_attn_step_x: bad argument"); return FMI_ERR_ARG; }
    const uint32_t waves = beams < 8 ? beams : 8;
    hipLaunchKernelGGL(k_cross_attn_step<float>, dim3(batch * heads), dim3(waves * 64), 0, (hipStream_t)stream, q_acc, ck, cv, bias, batch, beams, heads, S,
                       scale, out, q_bias, alpha, n_slabs, slab_stride, (__half *)out_planes, d_flag, pairs);
    NNCHK();
    return FMI_OK;
}
extern "C" int sealnn_cross_attn_step_x(void *stream, const float *q_acc, uint32_t n_slabs, uint64_t slab_stride, const float *q_bias, float alpha,
                                        const float *ck, const float *cv, const float *bias, uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S,
                                        float scale, float *out, void *out_planes, uint32_t *d_flag)
{ return cross_attn_step_x(stream, q_acc, n_slabs, slab_stride, q_bias, alpha, ck, cv, bias, batch, beams, heads, S, scale, out, out_planes, d_flag, 0); }
extern "C" int sealnn_cross_attn_step_x_pairs(void *stream, const float *q_acc, uint32_t n_slabs, uint64_t slab_stride, const float *q_bias, float alpha,
                                              const float *ck, const float *cv, const float *bias, uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S,
                                              float scale, float *out, void *out_planes, uint32_t *d_flag)
{ return cross_attn_step_x(stream, q_acc, n_slabs, slab_stride, q_bias, alpha, ck, cv, bias, batch, beams, heads, S, scale, out, out_planes, d_flag, 1); }

// ---- split GEMM operands (seal_amd/split_gemm.py): an fp32 row -> [hi | hi | lo'] in fp16, hi = fp16(x), lo' = fp16((x - hi) * 2^11) ----
// x = hi + lo' * 2^-11 to 22 bits; the GEMM  [hi | hi | lo'] . [W_hi | W_lo | W_hi * 2^-11]^T  on the fp16 matrix cores (fp32
// accumulate) is then x . W to fp32 accuracy (the dropped lo . lo term is 2^-22 of the product).  The lo plane is stored SCALED so that
// it has the magnitude of x itself: no fp16 subnormals whatever the matrix cores do with them.  A finite |x| beyond the fp16 range
// (65504) cannot be split: it is counted in *flag, which the caller checks (never seen in BART activations).
__global__ __launch_bounds__(256) void k_split_planes(const float *x, uint32_t rows, uint32_t K, __half *out, uint32_t *flag, uint32_t pairs)
{
    const uint32_t per_row = K / 4;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rows * per_row) return;
    const uint32_t r = (uint32_t)(i / per_row), c = (uint32_t)(i % per_row) * 4;
    const float4 v = *(const float4 *)(x + (uint64_t)r * K + c);
    const bool over = store_planes4(out + (uint64_t)r * (pairs ? 2 : 3) * K, K, c / 4, v, pairs);
    if (over && flag) atomicAdd(flag, 1u);
}

static int split_planes(void *stream, const float *x, uint32_t rows, uint32_t K, void *out, uint32_t *d_flag, uint32_t pairs)
{
    if (K % 4 || (pairs && K % 32)) { fmi_set_error("sealnn_split_planes: K=%u is not a multiple of %u", K, pairs ? 32 : 4); return FMI_ERR_UNSUPPORTED; }
    const uint64_t n = (uint64_t)rows * (K / 4);
    if (!n) return FMI_OK;
    hipLaunchKernelGGL(k_split_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows, K, (__half *)out, d_flag, pairs);
    NNCHK();
    return FMI_OK;
}
extern "C" int sealnn_split_planes(void *stream, const float *x, uint32_t rows, uint32_t K, void *out, uint32_t *d_flag)
{ return split_planes(stream, x, rows, K, out, d_flag, 0); }

// ---- the PAIRS forms: the same kernels writing their planes as [hi of 32 columns | lo * 2^11 of the same 32 columns] per 128-byte line, rows of 2 d halves:
// the operand of sealnn_hgemm_nt's PAIRS products (config bit 29; hgemm_kernels.hip), two thirds of the three-block operand ----
extern "C" int sealnn_split_planes_pairs(void *stream, const float *x, uint32_t rows, uint32_t K, void *out, uint32_t *d_flag)
{ return split_planes(stream, x, rows, K, out, d_flag, 1); }

extern "C" int sealnn_add_layernorm(void *stream, const float *x, const float *y, const float *gamma, const float *beta, uint32_t rows,
                                    uint32_t d, float eps, float *out)
{ return add_layernorm<float>(stream, x, y, gamma, beta, rows, d, eps, out); }
extern "C" int sealnn_add_layernorm_planes(void *stream, const float *x, const float *y, const float *gamma, const float *beta, uint32_t rows,
                                           uint32_t d, float eps, float *out, void *planes, uint32_t *d_flag)
{
    if (!planes) { fmi_set_error("sealnn_add_layernorm_planes: no plane buffer"); return FMI_ERR_ARG; }
    return add_layernorm<float>(stream, x, y, gamma, beta, rows, d, eps, out, planes, d_flag);
}
static int gelu_planes(void *stream, const float *x, uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag, const float *xb, float xa,
                       uint32_t n_slabs = 1, uint64_t slab_stride = 0, uint32_t pairs = 0)
{
    if (d % 4 || !planes || (pairs && d % 32)) { fmi_set_error("sealnn_gelu_planes: d=%u must be a multiple of 4 (and a plane buffer given)", d); return FMI_ERR_UNSUPPORTED; }
    const uint64_t n = (uint64_t)rows * (d / 4);
    if (!n) return FMI_OK;
    hipLaunchKernelGGL(k_gelu_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows, d, (__half *)planes, d_flag, xb, xa, n_slabs, slab_stride, pairs);
    NNCHK();
    return FMI_OK;
}
extern "C" int sealnn_gelu_planes(void *stream, const float *x, uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag)
{ return gelu_planes(stream, x, rows, d, planes, d_flag, nullptr, 1.f); }

// ---- the same kernels reading the RAW accumulators of a split GEMM: the projection's output is alpha * acc + bias (the GEMM's epilogue,
// which torch.addmm(out_dtype=float32) runs as a separate pass over the output: a copy of the broadcast bias in front of every product) ----
extern "C" int sealnn_self_attn_step_acc(void *stream, const float *qkv_acc, const float *qkv_bias, float alpha, float *kcache, float *vcache,
                                         const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out, int32_t *anc)
{
    if (!qkv_bias) { fmi_set_error("sealnn_self_attn_step_acc: no bias"); return FMI_ERR_ARG; }
    return self_attn_step<float>(stream, qkv_acc, kcache, vcache, d_t, rows, heads, T, scale, out, anc, qkv_bias, alpha);
}
extern "C" int sealnn_tree_self_attn_acc(void *stream, const float *qkv_acc, const float *qkv_bias, float alpha, const int32_t *anc, uint32_t n_nodes,
                                         uint32_t max_depth1, uint32_t heads, float scale, float *out)
{
    if (!qkv_bias) { fmi_set_error("sealnn_tree_self_attn_acc: no bias"); return FMI_ERR_ARG; }
    return tree_self_attn<float>(stream, qkv_acc, anc, n_nodes, max_depth1, heads, scale, out, qkv_bias, alpha);
}
extern "C" int sealnn_add_layernorm_acc(void *stream, const float *x, const float *y_acc, const float *y_bias, float alpha, const float *gamma,
                                        const float *beta, uint32_t rows, uint32_t d, float eps, float *out, void *planes, uint32_t *d_flag)
{
    if (!y_bias) { fmi_set_error("sealnn_add_layernorm_acc: no bias"); return FMI_ERR_ARG; }
    return add_layernorm<float>(stream, x, y_acc, gamma, beta, rows, d, eps, out, planes, d_flag, y_bias, alpha);
}
extern "C" int sealnn_add_layernorm_acc_slabs(void *stream, const float *x, const float *y_acc, uint32_t n_slabs, uint64_t slab_stride, const float *y_bias,
                                              float alpha, const float *gamma, const float *beta, uint32_t rows, uint32_t d, float eps, float *out,
                                              void *planes, uint32_t *d_flag)
{
    if (!y_bias) { fmi_set_error("sealnn_add_layernorm_acc_slabs: no bias"); return FMI_ERR_ARG; }
    if (n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_add_layernorm_acc_slabs: 1..16 slabs"); return FMI_ERR_ARG; }
    return add_layernorm<float>(stream, x, y_acc, gamma, beta, rows, d, eps, out, planes, d_flag, y_bias, alpha, n_slabs, slab_stride);
}

extern "C" int sealnn_gelu_planes_acc(void *stream, const float *x_acc, const float *x_bias, float alpha, uint32_t rows, uint32_t d, void *planes,
                                      uint32_t *d_flag)
{
    if (!x_bias) { fmi_set_error("sealnn_gelu_planes_acc: no bias"); return FMI_ERR_ARG; }
    return gelu_planes(stream, x_acc, rows, d, planes, d_flag, x_bias, alpha);
}
extern "C" int sealnn_gelu_planes_acc_slabs(void *stream, const float *x_acc, uint32_t n_slabs, uint64_t slab_stride, const float *x_bias, float alpha,
                                           uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag)
{
    if (!x_bias || n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_gelu_planes_acc_slabs: a bias and 1..16 slabs"); return FMI_ERR_ARG; }
    return gelu_planes(stream, x_acc, rows, d, planes, d_flag, x_bias, alpha, n_slabs, slab_stride);
}
extern "C" int sealnn_add_layernorm_acc_slabs_pairs(void *stream, const float *x, const float *y_acc, uint32_t n_slabs, uint64_t slab_stride, const float *y_bias,
                                                    float alpha, const float *gamma, const float *beta, uint32_t rows, uint32_t d, float eps, float *out,
                                                    void *planes, uint32_t *d_flag)
{
    if (!y_bias || !planes || n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_add_layernorm_acc_slabs_pairs: a bias, a plane buffer and 1..16 slabs"); return FMI_ERR_ARG; }
    return add_layernorm<float>(stream, x, y_acc, gamma, beta, rows, d, eps, out, planes, d_flag, y_bias, alpha, n_slabs, slab_stride, 1);
}
extern "C" int sealnn_gelu_planes_acc_slabs_pairs(void *stream, const float *x_acc, uint32_t n_slabs, uint64_t slab_stride, const float *x_bias, float alpha,
                                                  uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag)
{
    if (!x_bias || n_slabs < 1 || n_slabs > 16) { fmi_set_error("sealnn_gelu_planes_acc_slabs_pairs: a bias and 1..16 slabs"); return FMI_ERR_ARG; }
    return gelu_planes(stream, x_acc, rows, d, planes, d_flag, x_bias, alpha, n_slabs, slab_stride, 1);
}
extern "C" int sealnn_finish_product(void *stream, const float *acc, uint32_t n_slabs, uint64_t slab_stride, const float *bias, float alpha, uint32_t rows,
                                     uint32_t n, float *out)
{
    if (!acc || !bias || !out || n % 4 || n_slabs < 1 || n_slabs > 16 || (n_slabs > 1 && slab_stride % 4)) {
        fmi_set_error("sealnn_finish_product: n = %u must be a multiple of 4, 1..16 slabs, a bias", n);
        return FMI_ERR_ARG;
    }
    const uint64_t n4 = (uint64_t)rows * (n / 4);
    if (!n4) return FMI_OK;
    hipLaunchKernelGGL(k_finish_product, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, n_slabs, slab_stride, bias, alpha, n4, n / 4, out);
    NNCHK();
    return FMI_OK;
}
extern "C" int sealnn_add_layernorm_bf16(void *stream, const void *x, const void *y, const void *gamma, const void *beta, uint32_t rows,
                                         uint32_t d, float eps, void *out)
{ return add_layernorm<bf16>(stream, x, y, gamma, beta, rows, d, eps, out); }

extern "C" int sealnn_causal_self_attn(void *stream, const float *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale, float *out)
{ return causal_self_attn<float>(stream, qkv, n_seq, T, heads, scale, out); }
extern "C" int sealnn_causal_self_attn_bf16(void *stream, const void *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale, void *out)
{ return causal_self_attn<bf16>(stream, qkv, n_seq, T, heads, scale, out); }

extern "C" int sealnn_tree_self_attn(void *stream, const float *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t max_depth1, uint32_t heads,
                                     float scale, float *out)
{ return tree_self_attn<float>(stream, qkv, anc, n_nodes, max_depth1, heads, scale, out); }
extern "C" int sealnn_tree_self_attn_bf16(void *stream, const void *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t max_depth1, uint32_t heads,
                                          float scale, void *out)
{ return tree_self_attn<bf16>(stream, qkv, anc, n_nodes, max_depth1, heads, scale, out); }

extern "C" int sealnn_cross_attn_rows(void *stream, const float *q, const float *ck, const float *cv, const float *bias,
                                      const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S, float scale, float *out)
{ return cross_attn_rows<float>(stream, q, ck, cv, bias, row_batch, rows, heads, S, scale, out); }
extern "C" int sealnn_cross_attn_rows_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                                           const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S, float scale, void *out)
{ return cross_attn_rows<bf16>(stream, q, ck, cv, bias, row_batch, rows, heads, S, scale, out); }

extern "C" int sealnn_cross_attn_runs(void *stream, const float *q, const float *ck, const float *cv, const float *bias,
                                      const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads, uint32_t S, float scale,
                                      float *out)
{ return cross_attn_runs<float>(stream, q, ck, cv, bias, row_batch, rows, group, heads, S, scale, out); }
extern "C" int sealnn_cross_attn_runs_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                                           const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads, uint32_t S, float scale,
                                           void *out)
{ return cross_attn_runs<bf16>(stream, q, ck, cv, bias, row_batch, rows, group, heads, S, scale, out); }
