/*
 * sealfm.h -- C ABI of libsealfm.so, the MI355X (gfx950) FM-index engine that
 * replaces SEAL's sdsl-lite/SWIG extension `seal.cpp_modules._fm_index`.
 *
 * Every entry point cites the reference interface it replaces
 * (paths relative to /root/reference).  Conventions kept from the reference:
 *   - all rows / positions / symbols cross the boundary as uint64_t
 *     (reference: `unsigned long`, seal/cpp_modules/fm_index.hpp:16-18);
 *   - `backward_search_step` takes and returns the INCLUSIVE interval [l, r];
 *     `distinct*` take the HALF-OPEN interval [low, high);
 *     `backward_search_multi` returns (l, r+1);
 *   - symbols are already re-based (+SHIFT) on this side of the boundary, the
 *     Python layer adds/subtracts SHIFT (seal/index.py:16,109,141,154);
 *   - no exceptions: int status (0 = ok), `fmi_last_error()` for the text;
 *     sentinel results are replicated ((uint64_t)-1 from locate of a row >=
 *     size, (1,0) for a symbol outside the alphabet, empty result for low==high).
 *
 * Query entry points run on the GPU only.  There is no CPU query path in this
 * library: without a visible gfx950 device they return FMI_ERR_NO_DEVICE.
 * Host code here is limited to construction, (de)serialisation and upload.
 *
 * Two families of query entry points:
 *   fmi_<op>(...)      host buffers in/out (what the SWIG methods did, one call
 *                      per Python call); copies in, runs the kernels, copies out.
 *   fmi_dev_<op>(...)  device pointers + hipStream_t (passed as void*), no
 *                      synchronisation, no allocation; safe under hipGraph capture
 *                      once the workspace has been sized (fmi_dev_reserve).
 */
#ifndef SEALFM_H
#define SEALFM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fmi fmi_t;

enum {
    FMI_OK = 0,
    FMI_ERR_ARG = 1,
    FMI_ERR_IO = 2,
    FMI_ERR_NO_DEVICE = 3,
    FMI_ERR_HIP = 4,
    FMI_ERR_STATE = 5,
    FMI_ERR_UNSUPPORTED = 6,
    FMI_ERR_CAPACITY = 7
};

const char *fmi_last_error(void);
/* ABI version of this header; bumped on any signature change. */
uint32_t fmi_abi_version(void);
/* sha256 (64 hex digits) of the sources, headers and compiler flags this binary was built from; seal_amd/_lib.py refuses a library whose digest
 * is not that of the sources beside it (the binary is git-ignored and travels with snapshots of the tree) */
const char *fmi_source_digest(void);

/* ---- lifecycle / construction ------------------------------------------- */

/* FMIndex::FMIndex()  (fm_index.cpp:27-29) */
int fmi_create(fmi_t **out);
/* FMIndex::~FMIndex() (fm_index.cpp:31) */
void fmi_free(fmi_t *h);

/* FMIndex::initialize(const vector<char_type>&)  (fm_index.cpp:33-41).
 * `data` = n_data symbols, none of them 0; a 0 sentinel is appended as sdsl's
 * construct_im does, so fmi_size() == n_data + 1 afterwards.
 * Builds on the host (suffix array by prefix doubling) and uploads to `device`
 * (>= 0) or leaves the index host-only (device < 0; queries then fail). */
int fmi_build(fmi_t *h, const uint64_t *data, uint64_t n_data, int device);

/* FMIndex::initialize_from_file(file, width)  (fm_index.cpp:43-48): raw
 * little-endian integers of `width` bytes (1, 2, 4 or 8). */
int fmi_build_from_file(fmi_t *h, const char *path, int width, int device);

/* Same result as fmi_build, constructed on the GPU (suffix array by prefix
 * doubling with device radix sorts).  `d_data` is a DEVICE pointer to n_data
 * uint32 symbols.  Needed for corpora the host builder cannot reach
 * (NQ: ~2.9e9 symbols).  keep_host != 0 also downloads every array so that
 * fmi_save works. */
int fmi_build_device(fmi_t *h, const uint32_t *d_data, uint64_t n_data, int device, int keep_host);

/* Rank/select-only index from a BWT that is already on the device (exactly one 0 sentinel in it,
 * symbols `sym_bytes` = 2 or 4 bytes wide, all <= max_sym): wavelet matrix + tables, NO suffix array and
 * NO text, so locate / extract_text fail with FMI_ERR_STATE.  For the rank/select bandwidth stress tier
 * whose suffix array cannot be built on one GPU (SURVEY.md 8d tier X: ~1.4e10 symbols). */
int fmi_build_from_bwt_device(fmi_t *h, const void *d_bwt, uint64_t n, int sym_bytes, uint64_t max_sym, int device);

/* The same index as fmi_build_device for texts whose construction workspace there (prefix doubling: ~42 B per symbol) does not fit the
 * GPU next to the index -- BASELINE configs[4], 1.4e10 symbols: the suffix array is sorted in slices of ~slice_rows suffixes
 * (0: 2^30), cut by the value of their leading symbols and refined a few symbols at a time (seal_amd/csrc/fmi_build_gpu.hip,
 * build_sliced_impl; ~60 B of workspace per suffix of ONE slice).  d_text: the n symbols of the text in index order INCLUDING the
 * final 0 sentinel, sym_bytes (2 or 4) each, in device memory that the caller owns and keeps alive as long as the index lives: it
 * becomes the index's resident text, nothing is copied.  Replaces what sdsl's construct() does on disk for such sizes
 * (seal/cpp_modules/fm_index.cpp:43-48, seal/index.py:55-66). */
int fmi_build_device_sliced(fmi_t *h, const void *d_text, uint64_t n, int sym_bytes, int device, uint64_t slice_rows);

/* FMIndex::save(path)  (fm_index.cpp:186-189).  Own documented format
 * (DESIGN.md "on-disk layout"), not sdsl's. */
int fmi_save(const fmi_t *h, const char *path);
/* load_FMIndex(path)  (fm_index.cpp:191-199).  Accepts this engine's own container (fmi_save) and the files the
 * REFERENCE writes: sdsl-lite's serialisation of csa_wt_int<> (fm_index.cpp:186-189; the published SEAL indices,
 * README.md:67-69).  For the latter the text is recovered from the file (parallel LF walks from the ISA samples), the
 * index is rebuilt from it by the engine's builder, and the quirk table of DESIGN.md section 4 is taken from the file's
 * own bit layout (fmi_load_sdsl, seal_amd/csrc/fmi_sdsl.cpp; format restated from sdsl's published sources: unpinned
 * until a file written by sdsl itself has been read). */
int fmi_load(fmi_t **out, const char *path, int device);
int fmi_load_sdsl(fmi_t **out, const char *path, int device);

/* Upload a host-resident index (built with device < 0 or loaded) to a GPU. */
int fmi_to_device(fmi_t *h, int device);

/* Document boundaries `beginnings` (seal/index.py:25,50,59): n_docs+1 cumulative
 * token offsets; uploaded so that doc-id binning (index.py:77-82) runs on the GPU. */
int fmi_set_doc_beginnings(fmi_t *h, const uint64_t *beginnings, uint64_t n_entries);

/* A second handle on the same resident index: shares the device arrays of `src` (which must outlive the view) and
 * owns only the mutable per-pipeline state (workspace, incremental constraint state, counters, service stream), so that
 * several decode / retrieval pipelines can run on one GPU at a time, each on its own stream.  Free with fmi_free. */
int fmi_view_create(const fmi_t *src, fmi_t **out);

/* FMIndex::size()  (fm_index.cpp:50-52) */
uint64_t fmi_size(const fmi_t *h);
/* index.wavelet_tree.sigma (fm_index.cpp:38): distinct symbols incl. sentinel */
uint64_t fmi_sigma(const fmi_t *h);
uint64_t fmi_max_symbol(const fmi_t *h);
uint32_t fmi_levels(const fmi_t *h);
int fmi_device(const fmi_t *h);
/* bytes resident in HBM for this index */
uint64_t fmi_device_bytes(const fmi_t *h);

/* Host-side view of a built array, for tests of the host logic and for
 * hand-over to other tools.  name in {"sa","bwt","text","C","leaf","q1",
 * "dbase","sbase","wm"}; returns element count via *n_out and element size in bytes
 * via *elem_out; pointer stays owned by the index.  NULL if not host-resident. */
const void *fmi_host_array(const fmi_t *h, const char *name, uint64_t *n_out, uint32_t *elem_out);

/* ---- queries, host buffers (the SWIG surface) ---------------------------- */

/* FMIndex::backward_search_step(symbol, low, high)  (fm_index.cpp:67-76) */
int fmi_backward_search_step(fmi_t *h, uint64_t symbol, uint64_t low, uint64_t high, uint64_t out[2]);
/* FMIndex::backward_search_multi(query)  (fm_index.cpp:55-65) -> (l, r+1) */
int fmi_backward_search_multi(fmi_t *h, const uint64_t *query, uint64_t len, uint64_t out[2]);
/* batched form of the above over CSR sequences (offsets has n_seq+1 entries);
 * this is FMIndex.get_range (seal/index.py:102-111) for many sequences at once */
int fmi_backward_search_multi_batch(fmi_t *h, uint64_t n_seq, const uint64_t *offsets,
                                    const uint64_t *symbols, uint64_t *lo_out, uint64_t *hi_out);

/* FMIndex::distinct_count_multi(lows, highs)  (fm_index.cpp:111-131).
 * Results in CSR form: offsets_out[n+1]; symbols / counts ascending by symbol
 * per interval (sdsl interval_symbols order).  cap = capacity of syms_out /
 * cnts_out in entries; FMI_ERR_CAPACITY (with offsets_out filled so the caller
 * can size) if too small.  cnts_out may be NULL (FMIndex::distinct, cpp:78-89). */
int fmi_distinct_count_multi(fmi_t *h, uint64_t n, const uint64_t *lows, const uint64_t *highs,
                             uint64_t *offsets_out, uint64_t *syms_out, uint64_t *cnts_out, uint64_t cap);

/* FMIndex::locate(row)  (fm_index.cpp:163-167), batched; (uint64_t)-1 if row >= size.
 * doc_out (may be NULL) = bisect_right(beginnings, pos) - 1 (seal/index.py:77-82). */
int fmi_locate(fmi_t *h, uint64_t n, const uint64_t *rows, uint64_t *pos_out, uint64_t *doc_out);

/* FMIndex::extract_text(begin, end)  (fm_index.cpp:169-184): T[end-1] ... T[begin] */
int fmi_extract_text(fmi_t *h, uint64_t begin, uint64_t end, uint64_t *out);

/* ---- queries, device pointers + stream (the decode / retrieval hot path) -- */

/* workspace for the fmi_dev_* calls: rows = max intervals per call */
int fmi_dev_reserve(fmi_t *h, uint64_t max_rows);

/* backward_search_step for n independent (symbol, [l, r]) triples */
int fmi_dev_bs_step(fmi_t *h, void *stream, uint64_t n, const uint64_t *d_sym, const uint64_t *d_lo,
                    const uint64_t *d_hi, uint64_t *d_lo_out, uint64_t *d_hi_out);

/* get_range over CSR sequences of RAW token ids (+shift applied here) */
int fmi_dev_get_range(fmi_t *h, void *stream, uint64_t n_seq, const int64_t *d_offsets,
                      const int64_t *d_tokens, int64_t shift, uint64_t *d_lo_out, uint64_t *d_hi_out);

/* IndexBasedLogitsProcessor.__call__ for cur_len >= 2 (seal/beam_search.py:79-140):
 *   d_input_ids  int64 [rows, cur_len] row-major (decoder ids incl. the start token)
 *   d_scores_in  fp32 [rows, vocab]; d_scores_out fp32 [rows, vocab] (may alias)
 *   force_from   host array of n_force raw token ids (force_decoding_from) or NULL
 * out[r][v] = in[r][v] if v is an allowed continuation of row r else -inf. */
int fmi_dev_constrain_scores(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len,
                             const int64_t *d_input_ids, const float *d_scores_in, float *d_scores_out,
                             uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                             const int64_t *force_from, uint64_t n_force,
                             int64_t stop_at_count, int always_allow_eos);

/* Same constraint as a bitmap: d_bits uint32 [rows, ceil(vocab/32)], bit v of row
 * r set iff v allowed.  Building block of the fused masked top-k. */
int fmi_dev_allowed_bits(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len,
                         const int64_t *d_input_ids, uint32_t *d_bits,
                         uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                         const int64_t *force_from, uint64_t n_force,
                         int64_t stop_at_count, int always_allow_eos);

/* One decode step of constrained_beam_search fused (seal/beam_search.py:244-310): log_softmax +
 * InfNanRemove of d_logits [batch*beams, vocab], + beam score, constraint (as fmi_dev_allowed_bits; for
 * cur_len == 1 the caller's d_first_bits bitmap of occurring_distinct, with always_allow_eos already folded
 * in), top-(2*beams) per query on the CONSTRAINED scores; outputs [batch, 2*beams]: flat index
 * (beam * vocab + token), constrained score, UNCONSTRAINED score (the value the reference carries on).
 * Ties go to the lower flat index; when a query has fewer finite candidates the rest is filled with
 * not-allowed tokens (constrained -inf), which torch.topk leaves unspecified.  d_scratch: at least
 * 4 * batch*beams * (3 + 4*beams) bytes.  Nothing of shape [rows, vocab] is written. */
int fmi_dev_constrained_topk(fmi_t *h, void *stream, uint64_t batch, uint64_t beams, uint64_t cur_len,
                             const int64_t *d_input_ids, const float *d_logits, const float *d_beam_scores,
                             uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id, const int64_t *force_from,
                             uint64_t n_force, int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                             void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc);

/* The same step inside a decode LOOP that can vouch for its own continuity: `state_tag` (non-zero,
 * the same for every step of one generate call) + `d_parent_rows[rows]` = for every row the row of
 * the PREVIOUS call (same tag, cur_len - 1) whose sequence it extends by one token (the beam_idx of
 * beam_search.py:661).  The index then keeps every row's prefix range between calls and advances it
 * by ONE backward-search step instead of re-searching the whole prefix as the reference does
 * (beam_search.py:87-105) -- same ranges, bit for bit.  Any break in the chain (other tag, other row
 * count, cur_len not previous + 1, cur_len == 1) falls back to the full search.  state_tag == 0:
 * identical to fmi_dev_constrained_topk.  One decode loop per index handle at a time. */
int fmi_dev_constrained_topk_step(fmi_t *h, void *stream, uint64_t batch, uint64_t beams, uint64_t cur_len,
                                  const int64_t *d_input_ids, const float *d_logits, const float *d_beam_scores,
                                  uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id, const int64_t *force_from,
                                  uint64_t n_force, int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                                  void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                                  uint64_t state_tag, const int64_t *d_parent_rows);

/* The same step for SEVERAL decodes that advance in lockstep as one loop (the searcher's body decode,
 * retrieval.py:70-83, and title decode, retrieval.py:162-176, of one batch of queries: their rows stacked,
 * 2 * batch * beams rows per model step instead of two loops of batch * beams): group g = the next
 * group_batch[g] queries (x beams rows) with its own end-of-sequence token group_eos[g] and forced prefix
 * (group_force[g * 8 ..], group_n_force[g] tokens; with ONE group: group_force[0 ..]).  1 <= n_groups <= 3;
 * pad / always_allow_eos are shared; group_stop_at_count[g] (NULL: stop_at_count for every group) is the group's own
 * stop_at_count -- the reference passes it to the body decode only (retrieval.py:70-83; titles / codes run with 0,
 * retrieval.py:162-176, 212-236).  Every row's mask is the one fmi_dev_constrained_topk_step
 * computes for it with its group's arguments, and every query's top-2K likewise: the call is ONE constraint
 * launch over all rows.  A later call of the same loop may hold fewer rows (finished decodes dropped):
 * d_parent_rows[] then still names rows of the previous call. */
int fmi_dev_constrained_topk_groups(fmi_t *h, void *stream, uint64_t n_groups, const uint64_t *group_batch,
                                    const int64_t *group_eos, const int64_t *group_force, const uint64_t *group_n_force,
                                    uint64_t beams, uint64_t cur_len, const int64_t *d_input_ids, const float *d_logits,
                                    const float *d_beam_scores, uint64_t vocab, int64_t shift, int64_t pad_id,
                                    int64_t stop_at_count, int always_allow_eos, const uint32_t *d_first_bits,
                                    void *d_scratch, uint64_t scratch_bytes, int64_t *d_top_idx, float *d_top_con, float *d_top_unc,
                                    uint64_t state_tag, const int64_t *d_parent_rows, const int64_t *group_stop_at_count);

/* fmi_dev_allowed_bits with the continuity contract of fmi_dev_constrained_topk_step (state_tag / d_parent_rows:
 * the rows extend, by one token, rows d_parent_rows[] of the previous call with the same tag, so the prefix range
 * advances by ONE backward-search step).  d_bits == NULL: the bitmap is written to the index's own workspace
 * (two buffers alternating between calls, each call clearing the other: no memset launch) and *d_bits_out
 * points at it, valid until the next constraint call but one on this handle. */
int fmi_dev_allowed_bits_step(fmi_t *h, void *stream, uint64_t rows, uint64_t cur_len, const int64_t *d_input_ids,
                              uint32_t *d_bits, uint64_t vocab, int64_t shift, int64_t pad_id, int64_t eos_id,
                              const int64_t *force_from, uint64_t n_force, int64_t stop_at_count, int always_allow_eos,
                              uint64_t state_tag, const int64_t *d_parent_rows, const uint32_t **d_bits_out);

/* tools only: while d_buf (n_words uint64, >= 8 per wave of the constraint launch) is set, every wave of
 * k_constrain stores realtime stamps (100 MHz) at {start, prefix range known, root child known (0: none),
 * sub-tree expanded, bitmap stored}.  NULL switches it off. */
int fmi_dev_debug_timestamps(fmi_t *h, uint64_t *d_buf, uint64_t n_words);

/* Launch-shape options of the constraint / top-2K / aggregation calls, for A/B measurements and for the tests that force every kernel path.
 * A handle starts with the built-in choices (nothing is read from the environment: round 6); this call changes one, value -1 restores the
 * built-in choice.  Names: "constrain_waves" (1: one self-contained wave per (row, top digit)),
 * "leave_early" (0: the waves of empty items stay), "row_first" (0 / 1: never / always the row-first pair of launches),
 * "row_first_from" (prefix length in tokens from which a call goes row-first),
 * "prefix_tables" (0: the first constrained step of a decode through the generic expansion instead of the per-token node tables),
 * "table_grid" (workgroups of the table call's flat pass),
 * "topk_narrow" (rows with more allowed tokens take the wide-row path of the top-2K kernel), "topk_legacy" (1: exact radix
 * select on wide rows), "chain_steps" (0: fmi_dev_beam_step leaves the rows' chains to the next constraint call),
 * "advance_apart" (measurement passes: 0 = k_beam_advance as the product's one launch, timed whole), "agg_rank_by_sorts"
 * (fmi_dev_aggregate's first-stage ranking: 1 = the three full stable sorts of rounds 2-5, 2 = the single-workgroup selection; 0 = the
 * selection on twelve workgroups per query), "pt_inject_failure" (tests: building a prefix table fails).  Results are identical for every setting (tests/test_gpu_fmindex.py, tests/test_gpu_decode.py).
 *
 * "leave_early" = 1 (the default): a wave of k_constrain whose (row, top digit) item is empty ENDS before its workgroup's barriers.
 * That relies on the documented behaviour of the gfx9 / CDNA barrier -- "S_BARRIER: Synchronize waves within a threadgroup. [...] If
 * some waves in the threadgroup have already terminated, this waits on only the surviving waves" (AMD Instinct MI300 / CDNA3 ISA
 * reference guide, SOPP instructions, S_BARRIER; the barrier counts a workgroup's waves that have not executed s_endpgm) -- not on
 * the HIP programming model, which leaves __syncthreads() in divergent code undefined: the kernels are gfx950-only, the condition is
 * wave-uniform (a scalar branch: a whole wave leaves or stays), and the GPU tests run every constraint form with 0 and 1. */
int fmi_dev_set_option(fmi_t *h, const char *name, int64_t value);

/* The per-token node tables this handle has built so far (one per forced prefix of a decode; the first constrained step of a decode
 * reads them instead of expanding from the root): how many, their leaf-level nodes in total, the HBM they hold. */
int fmi_dev_prefix_table_stats(fmi_t *h, uint64_t *n_tables, uint64_t *n_nodes, uint64_t *n_bytes);

/* tools only (tools/soak.py: which launch of a stalled stream never completed): while `marks` (>= 16 uint32 of
 * host-visible memory, e.g. pinned) is set, fmi_dev_aggregate writes marks[1] = 100 * call number + stage after every
 * launch of its sequence, in stream order.  NULL switches it off.  fmi_dev_mark: one such write on any stream. */
int fmi_dev_debug_marks(fmi_t *h, uint32_t *marks);
int fmi_dev_mark(void *stream, uint32_t *word, uint32_t value);

/* locate + doc binning for n rows (seal/keys.py:320-324) */
int fmi_dev_locate(fmi_t *h, void *stream, uint64_t n, const uint64_t *d_rows,
                   uint64_t *d_pos_out, uint64_t *d_doc_out);

/* rows of many half-open ranges, each truncated to `max_per_range`
 * (islice(range(*get_range(ngram)), max_hits), seal/keys.py:320), located and
 * binned in one launch.  d_out_offsets[n_ranges+1] is an INPUT (exclusive scan of
 * min(hi-lo, max_per_range)), outputs are written at those offsets. */
int fmi_dev_locate_ranges(fmi_t *h, void *stream, uint64_t n_ranges, const uint64_t *d_lo,
                          const uint64_t *d_hi, uint64_t max_per_range, const uint64_t *d_out_offsets,
                          uint64_t total, uint64_t *d_pos_out, uint64_t *d_doc_out);

/* get_doc for many documents (seal/index.py:68-75): forward-order RAW token ids
 * (symbol - shift) of documents d_docs[i] written at d_out + d_out_offsets[i] */
int fmi_dev_get_docs(fmi_t *h, void *stream, uint64_t n_docs, const uint64_t *d_docs,
                     const uint64_t *d_out_offsets, int64_t shift, int64_t *d_out);

/* probe counter of the fmi_dev_* constraint / expansion launches since the last
 * read: the distinct 128-byte wavelet-matrix blocks the rank probes loaded (x 128 =
 * DESIGN.md "algorithmic bytes").  Device-side counter, read back synchronously:
 * for measurement only.
 * fmi_dev_read_expand_stats: out4 = {blocks, wave iterations of the sub-tree
 * expansion, nodes expanded, nodes a binary wavelet tree over the same symbols
 * visits} without resetting (nodes / (32 * iterations) = lane-pair utilisation of
 * k_constrain: one node per pair of lanes; 2 * out4[3] = the 64-byte level-probes
 * of the reference-shaped algorithm for the same work). */
int fmi_dev_enable_probe_count(fmi_t *h, int enable);
int fmi_dev_read_probe_count(fmi_t *h, uint64_t *probes_out);
int fmi_dev_read_expand_stats(fmi_t *h, uint64_t *out4);

/* HIP-event timing of the constraint kernel (k_constrain): when enabled, every launch
 * is bracketed by two events recorded on the launch stream.  read = synchronise,
 * sum the elapsed times since the last read, reset.  For bench.py's roofline. */
int fmi_dev_enable_timing(fmi_t *h, int enable);
int fmi_dev_read_timing(fmi_t *h, uint64_t *launches_out, double *total_ms_out);

/* ---- one decode step of the beam loop behind the model's forward -----------------------------------------------
 * reference seal/beam_search.py:244-332 (log-softmax, processors, top-2K on the constrained scores carrying the unconstrained
 * ones, next_indices / next_tokens) + BeamSearchScorerWithMemory.process (:614-700: every ranked candidate recorded, the
 * first num_beams non-eos candidates continue) + input_ids = cat(input_ids[beam_idx], beam_next_tokens) + _reorder_cache,
 * for the stacked rows of up to three decodes in lockstep (group g = the next group_batch[g] queries: own eos, forced prefix,
 * stop_at_count), as fmi_dev_constrained_topk_groups + ONE more launch (k_beam_advance) that also runs, for every NEW row,
 * the dependent chain of the NEXT step's constraint (kept range of its source row -> one backward-search step with its token
 * -> class -> root node split; `chain_next`), so that the next call of the same `state_tag` starts at the sub-trees.
 * All pointers are device pointers; nothing is allocated or synchronised.
 *   d_ids          [rows][ids_stride] int64: the rows' tokens, cur_len of each used; rewritten in place to cur_len + 1
 *   d_beam_scores  [rows] in: of the rows; out: of the new rows        d_beam_idx [rows] out: source row of every new row
 *   d_tokens_out   [rows] or null: the new rows' last tokens (the decoder's next input)
 *   d_anc          null, or the decoder's ancestry table [anc_positions][anc_rows] int32: column r' := column source(r')
 *   d_hist_tok[g]  null, or [group_batch[g]][hist_H[g]][hist_L[g]] int64 (-1-filled by the caller): candidate c of the step is
 *                  written at hypothesis slot hist_off + c (its source row's tokens, then its token); d_hist_sc[g] [batch][H] f32
 *   d_top_*        [batch][2 * beams] scratch outputs of the merge (kept: tests)
 *   dropped_rows   rows that left the FRONT of the loop since the previous step of this state_tag (a decode that ended)      */
typedef struct fmi_beam_step {
    uint64_t struct_bytes;           /* sizeof(fmi_beam_step_t) */
    uint64_t n_groups;
    uint64_t group_batch[3];
    int64_t group_eos[3];
    int64_t group_force[3][8];
    uint64_t group_n_force[3];
    int64_t group_stop[3];
    uint64_t beams, cur_len, vocab;
    int64_t shift, pad_id;
    int32_t always_allow_eos, chain_next;
    int64_t *d_ids;
    uint64_t ids_stride;
    const float *d_logits;
    float *d_beam_scores;
    const uint32_t *d_first_bits;
    void *d_scratch;
    uint64_t scratch_bytes;
    int64_t *d_top_idx;
    float *d_top_con, *d_top_unc;
    int64_t *d_beam_idx, *d_tokens_out;
    int32_t *d_anc;
    uint64_t anc_rows, anc_positions;
    int64_t *d_hist_tok[3];
    float *d_hist_sc[3];
    uint64_t hist_H[3], hist_L[3], hist_off;
    uint64_t state_tag, dropped_rows;
} fmi_beam_step_t;
int fmi_dev_beam_step(fmi_t *h, void *stream, const fmi_beam_step_t *step);
/* the allowed-token bitmap the last constraint call of this handle filled ([rows][words_per_row] uint32, device memory owned by the
 * handle, valid until the next call; null after a cur_len == 1 step): tests and bench.py's parity check read the masks that were
 * actually applied (table / chained / generic form alike) instead of recomputing them */
const uint32_t *fmi_dev_last_constraint_bits(fmi_t *h, uint64_t *rows_out, uint64_t *words_per_row_out);

/* Per-call log of the constraint calls (bench.py's `roofline.by_call`): while enabled, every call of the
 * fmi_dev_allowed_bits* / fmi_dev_constrain_scores / fmi_dev_constrained_topk* family appends one record -- prefix length
 * (cur_len), rows, launch form (FMI_CALL_*) -- which names its HIP-event pair when fmi_dev_enable_timing is on, and, when
 * fmi_dev_enable_probe_count is on, holds the 128-byte blocks its launches loaded (the stream is drained after the call:
 * a measurement pass).  read = synchronise, copy up to `cap` records out (us = -1 without timing), report how many there
 * were, clear.  The reference has no counterpart: one IndexBasedLogitsProcessor.__call__ (beam_search.py:62-140) = one record. */
#define FMI_CALL_GENERIC 0   /* k_constrain alone */
#define FMI_CALL_ROW_FIRST 1 /* k_constrain_rows + k_constrain */
#define FMI_CALL_TABLE 2     /* k_constrain_table + k_table_bits (first constrained step of a decode) */
#define FMI_CALL_CHAINED 3   /* k_constrain alone, started from what the previous step's k_beam_advance left (fmi_dev_beam_step) */
#define FMI_CALL_ADVANCE 4   /* k_beam_advance without / with (5) the chains of the call it precedes; cur_len = that call's */
#define FMI_CALL_ADVANCE_CHAIN 5
int fmi_dev_call_log(fmi_t *h, int enable);
int fmi_dev_read_call_log(fmi_t *h, uint64_t cap, uint32_t *cur_len, uint32_t *rows, uint32_t *kind, float *us, uint64_t *blocks,
                          uint64_t *n_out);

/* Stage timing of fmi_dev_aggregate (bench.py's `roofline_aggregate`; a measurement pass: every call then ends with a stream
 * synchronise and a read-back of what it processed).  While enabled every call (up to 64) brackets its stages with HIP events:
 *   0 k_agg_locate (row -> text position -> document: keys.py:322-324, index.py:77-82)   1 radix sort by (query, position)
 *   2 coverage (k_mis_prepare + k_mis: keys.py:320-341)   3 k_doc_keys + radix sort by (query, document)
 *   4 entry boundaries (k_heads, scan, k_entry_starts)     5 k_entries (keys.py:343-364)
 *   6 ranking (three stable radix sorts + k_top_docs = sorted(first_stage.items()), keys.py:366-375)
 *   7 per-query token tables (memsets + scatters)          8 k_full_score over the ranked documents (keys.py:377-497)
 *   9 k_rank_docs                                          10 k_full_score over the caller's top-k (the records that go back)
 * read: stage_ms[11] summed over the calls; counts[5] = {located rows, (query, document) entries, documents scored in stage 8,
 * documents re-scored in stage 10, tokens of the documents of stage 8}; clears. */
#define FMI_AGG_STAGES 11
int fmi_dev_agg_timing(fmi_t *h, int enable);
int fmi_dev_read_agg_timing(fmi_t *h, double *stage_ms, uint64_t *counts, uint64_t *calls_out);

/* A copy between device memory and PINNED (device-visible) host memory, either way, done by a kernel: ordered by `stream` alone, where a
 * hipMemcpyAsync waits in the DMA engine's queue behind copies other streams have enqueued (the plan and the results of
 * fmi_dev_aggregate behind the pending copy-back of a decode enqueued ahead: seal_amd/csrc/fmi_upload.hip).  dst, src and bytes are
 * multiples of 4 (16: wider accesses); the host sees a download once the stream has been waited for. */
int fmi_dev_kernel_copy(void *stream, void *dst, const void *src, uint64_t bytes);

/* Device pointer of a resident array, for zero-copy hand-over (e.g. to build the
 * CPU baseline's samples).  name in {"sa_lo","sa_hi","text","wm","C","leaf","q1","doc_begin"}. */
const void *fmi_dev_array(const fmi_t *h, const char *name, uint64_t *n_out, uint32_t *elem_out);

/* ---- first-stage evidence aggregation (host, consumes fmi_dev_locate_ranges output) --
 * seal/keys.py:311-367 for one query: keys in processing order (descending score),
 * CSR tokens per key, score per key, CSR of located rows per key with their text
 * position and document; coverage window [pos - len, pos) as in the reference.
 * Result = documents in ranked order (stable sort of first-touch order by
 * (1-single_key)*(-score) + single_key*(-best_score), cut to n_top), each with its
 * re-weighted score, best key, and the (key index, re-weighted score) list. */
typedef struct fmi_evidence fmi_evidence_t;
int fmi_first_stage(uint64_t n_keys, const int64_t *key_tok_off, const int64_t *key_toks, const double *key_score,
                    const int64_t *occ_off, const int64_t *pos, const int64_t *doc, int allow_overlaps,
                    double beta, double single_key, uint64_t n_top, fmi_evidence_t **out);
uint64_t fmi_evidence_docs(const fmi_evidence_t *ev);
uint64_t fmi_evidence_entries(const fmi_evidence_t *ev);
int fmi_evidence_read(const fmi_evidence_t *ev, int64_t *doc, double *score, int64_t *best_key, double *best_score,
                      int64_t *key_off, int32_t *key_idx, double *key_score);
void fmi_evidence_free(fmi_evidence_t *ev);

/* ---- full-document scoring (host) -- seal/keys.py:377-494 for the ranked documents of one query:
 * keys with score > 0 (CSR tokens, in all_ngrams order), optional per-token unigram scores [vocab],
 * documents as CSR token arrays (already in the `[2] + doc[:-1]` form of keys.py:388).  Result =
 * documents sorted by descending score (stable), each with score, best single key, and the list of
 * accepted (key index | -(token+1) for a unigram, discounted score) in acceptance order. */
typedef struct fmi_fullscore fmi_fullscore_t;
int fmi_full_score(uint64_t n_keys, const int64_t *key_tok_off, const int64_t *key_toks, const double *key_score,
                   const double *type_scores, uint64_t vocab, uint64_t n_docs, const int64_t *doc_off,
                   const int64_t *doc_toks, int allow_overlaps, double beta, double single_key,
                   int single_key_add_unigrams, int unigrams_ignore_free_places, fmi_fullscore_t **out);
uint64_t fmi_fullscore_docs(const fmi_fullscore_t *fs);
uint64_t fmi_fullscore_entries(const fmi_fullscore_t *fs);
int fmi_fullscore_read(const fmi_fullscore_t *fs, int64_t *order, double *score, int64_t *best_key, double *best_score,
                       int64_t *pick_off, int64_t *pick_id, double *pick_score);
void fmi_fullscore_free(fmi_fullscore_t *fs);

/* ---- evidence aggregation on the GPU: seal/keys.py:311-497 (first stage + full-document scoring) for a chunk of
 * queries, from the scored keys to the ranked documents, without the located rows, the candidate documents or their
 * text ever leaving the device.  Replaces what fmi_first_stage + fmi_full_score compute on the host (those two stay
 * as the bit-exact checkers); same results, same order, same float64 arithmetic.
 *
 * fmi_agg_pack (host): the "table keys" of every query = the keys of `all_ngrams` with score > 0 (keys.py:305-309,
 * 377-381) in all_ngrams order (descending score, stable), CSR over the queries; key_rare marks the keys of
 * `rare_ngrams` (keys.py:285-299), whose rows [key_lo, key_hi) -- cut to max_hits -- the first stage locates;
 * type_scores[q] = the query's `unigram_scores` after keys.py:236-272 (dense [vocab], or NULL).  The plan owns one
 * packed blob (layout: seal_amd/csrc/fmi_agg.h) that the caller copies to the GPU as it is. */
typedef struct fmi_agg_plan fmi_agg_plan_t;
int fmi_agg_pack(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                 const double *key_score, const uint8_t *key_rare, const uint64_t *key_lo, const uint64_t *key_hi,
                 uint64_t max_hits, uint64_t index_size, const double *const *type_scores, uint64_t vocab,
                 fmi_agg_plan_t **out);
const void *fmi_agg_plan_blob(const fmi_agg_plan_t *plan, uint64_t *bytes_out);
uint64_t fmi_agg_plan_occurrences(const fmi_agg_plan_t *plan);
void fmi_agg_plan_free(fmi_agg_plan_t *plan);

/* Key scoring of aggregate_evidence (keys.py:207-309) + fmi_agg_pack in one call, for a chunk of queries: input = the keys
 * as the searcher hands them over (CSR tokens, LM log-probabilities, row ranges from one backward-search launch), the
 * model's unigram log-probabilities per query ([vocab] each, or NULL), the per-index single-token range table, and
 * aggregate_evidence's parameters; float64 through libm (log, exp, pow) in the reference's operation order.  The plan
 * additionally holds `all_ngrams` of every query (fmi_agg_plan_ngrams: src = index into the query's input keys, or
 * -(token+1) for a unigram added by keys.py:274-278; score; rare flag) and, per table key, its src (fmi_agg_plan_table_src). */
int fmi_agg_score_pack(uint64_t n_queries, const int64_t *q_key_off, const int64_t *key_tok_off, const int64_t *key_toks,
                       const double *key_lm_score, const uint64_t *key_lo, const uint64_t *key_hi,
                       const double *const *unigram_logprobs, uint64_t vocab, const int64_t *uni_lo, const int64_t *uni_hi,
                       uint64_t n_uni_table, double ntokens, double alpha, double length_penalty, double smoothing,
                       int use_fm_index_frequency, int add_best_unigrams_to_ngrams, int64_t use_top_k_unigrams,
                       uint64_t max_occurrences_1, uint64_t max_occurrences_2, uint64_t index_size, fmi_agg_plan_t **out);
uint64_t fmi_agg_plan_ngrams(const fmi_agg_plan_t *plan, uint64_t query, int64_t *src, double *score, uint8_t *rare);
uint64_t fmi_agg_plan_table_src(const fmi_agg_plan_t *plan, int64_t *src);

/* byte offsets into the output buffer of fmi_dev_aggregate, as fmi_dev_aggregate_sizes reports them in out_layout[20].
 * R = n_queries * keep records, record (q, x) = x-th best document of query q at index q * keep + x. */
enum {
    FMI_AGG_OUT_N_OUT = 0,        /* u32 [nq]   documents returned for the query (<= keep) */
    FMI_AGG_OUT_FLAGS = 1,        /* u32 [nq]   bit 0: the query exceeded a device limit -> recompute it with the host checkers */
    FMI_AGG_OUT_CURSOR = 2,       /* u32 [2]    used entries of the pick pool / the token pool */
    FMI_AGG_OUT_REC_DOC = 3,      /* u64 [R]    document index */
    FMI_AGG_OUT_REC_SCORE = 4,    /* f64 [R]    keys.py:493 */
    FMI_AGG_OUT_REC_BEST_SCORE = 5, /* f64 [R]  score of the best single key (keys.py:424-441) */
    FMI_AGG_OUT_REC_BEST_KEY = 6, /* i32 [R]    its table key id, -1 if no key occurs in the document */
    FMI_AGG_OUT_REC_T = 7,        /* u32 [R]    tokens of the document */
    FMI_AGG_OUT_REC_NPICKS = 8,   /* u32 [R]    accepted keys + unigrams (keys.py:466,487) */
    FMI_AGG_OUT_REC_PICK_OFF = 9, /* u32 [R]    first pick of the record in the pick pool */
    FMI_AGG_OUT_REC_TOK_OFF = 10, /* u32 [R]    first token of the record in the token pool */
    FMI_AGG_OUT_FS_CNT = 11,      /* u32 [nq]   documents of the first-stage ranking (<= n_top) */
    FMI_AGG_OUT_PICK_ID = 12,     /* i32 pool   table key id, or -(token + 1) for a unigram */
    FMI_AGG_OUT_PICK_SCORE = 13,  /* f64 pool   its discounted score */
    FMI_AGG_OUT_TOKENS = 14,      /* i32 pool   document tokens as scored: [2] + get_doc(doc)[:-1] (keys.py:388) */
    FMI_AGG_OUT_FS_DOC = 15,      /* u32 [nq][n_top]  first-stage ranking (keys.py:366) */
    FMI_AGG_OUT_FS_SCORE = 16,    /* f64 [nq][n_top]  its scores after the repetition discount (keys.py:352-364) */
    FMI_AGG_OUT_FIXED_BYTES = 17, /* arrays 0..11 live in [0, fixed_bytes): one small copy tells how much of the pools to fetch */
    FMI_AGG_OUT_BYTES = 18
};
int fmi_dev_aggregate_sizes(fmi_t *h, const fmi_agg_plan_t *plan, uint64_t n_top, uint64_t keep, int allow_overlaps,
                            uint64_t *ws_bytes, uint64_t *out_layout /* [20] */);

/* the whole aggregation on `stream`: d_plan_blob = the plan's blob copied to the device, d_ws / d_out = device buffers of
 * the sizes reported above.  n_top = n_docs_complete_score (documents that are fully scored), keep = documents recorded
 * per query (the caller's k); the other parameters are aggregate_evidence's (keys.py:178-204).  Asynchronous. */
int fmi_dev_aggregate(fmi_t *h, void *stream, const fmi_agg_plan_t *plan, const void *d_plan_blob, uint64_t n_top, uint64_t keep,
                      int allow_overlaps, double beta, double single_key, int single_key_add_unigrams,
                      int unigrams_ignore_free_places, int64_t shift, void *d_ws, uint64_t ws_bytes, void *d_out, uint64_t out_bytes);

/* (sr + log(1-exp(snr))) - (snr + log(1-exp(sr))), snr = log((count+smoothing)/(ntokens+smoothing)), 0 where
 * count == 0 -- seal/keys.py:221-224,258-261 for n pairs, libm doubles (what python's math module calls). */
int fmi_log_odds_batch(uint64_t n, const double *sr, const int64_t *count, double ntokens, double smoothing, double *out);

#ifdef __cplusplus
}
#endif
#endif /* SEALFM_H */
