// Evidence aggregation on the GPU (reference seal/keys.py:311-497): layout of the packed key plan that the host
// builds per chunk of queries (fmi_agg_pack.cpp) and the kernels consume (fmi_aggregate.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static constexpr uint64_t FMI_AGG_MAGIC = 0x31474741494d4653ull;   // "SFMIAGG1"
static constexpr uint32_t FMI_AGG_MAX_QUERIES = 256;               // query id takes 8 bits of the sort keys
static constexpr uint32_t FMI_AGG_MAX_KEY_LEN = 255;               // text positions are offset by 256 in the sort key
static constexpr uint32_t FMI_AGG_POS_BITS = 41;                   // (pos + 256) < 2^41 for texts below 2^40 symbols
static constexpr uint32_t FMI_AGG_TRIE_EMPTY = 0xFFFFFFFFu;

// One blob, offsets in bytes from its start, every array 16-byte aligned.  "table keys" = the keys of all_ngrams
// with score > 0 (keys.py:305-309, 377-381), per query in all_ngrams order (descending score, stable); the rare keys
// (keys.py:285-299) are the sub-sequence of them that the first stage locates, in the same order.
struct FmiAggHeader {
    uint64_t magic, bytes;
    uint64_t nq, n_keys, n_rare, total_occ, vocab, max_key_len, max_u, max_q_keys, n_uni, n_trie_slots, n_tok;
    uint64_t o_q_key_off;     // u32 [nq+1]      table keys of query q = [q_key_off[q], q_key_off[q+1])
    uint64_t o_q_rare_off;    // u32 [nq+1]      rare keys of query q (indices into rare_key)
    uint64_t o_rare_key;      // u32 [n_rare]    table key id of the r-th rare key (processing order)
    uint64_t o_rare_occ_off;  // u64 [n_rare+1]  first occurrence index of rare key r; occurrences = min(count, max_hits)
    uint64_t o_key_lo;        // u64 [n_keys]    first row of the key's range
    uint64_t o_key_len;       // u32 [n_keys]    tokens in the key
    uint64_t o_key_q;         // u32 [n_keys]    query of the key
    uint64_t o_key_rank;      // u32 [n_keys]    rank within its query by (-score, token sequence): the heap order of keys.py:431
    uint64_t o_key_score;     // f64 [n_keys]
    uint64_t o_kset_off;      // u32 [n_keys+1]  distinct tokens of a key as query-local ids (repetition(), keys.py:186-191)
    uint64_t o_kset_ids;      // u32 [kset_off[n_keys]]
    uint64_t o_q_tok_off;     // u32 [nq+1]      query-local token ids: local id x of query q is token tok_list[q_tok_off[q] + x]
    uint64_t o_tok_list;      // u32 [n_tok]
    uint64_t o_q_trie_off;    // u32 [nq+1]      slots of query q's trie hash table (a power of two)
    uint64_t o_trie;          // uint4 [n_trie_slots]  {parent node, token, child node, key at child or 0xFFFFFFFF}; parent = EMPTY: free slot
    uint64_t o_uni_flat;      // u64 [n_uni]     q * vocab + token of a non-zero unigram score (keys.py:236-272)
    uint64_t o_uni_score;     // f64 [n_uni]
};

__host__ __device__ static inline uint32_t fmi_agg_trie_hash(uint32_t node, uint32_t tok)
{
    uint32_t h = node * 0x9E3779B1u ^ tok * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}
