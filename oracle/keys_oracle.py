"""TEST INFRASTRUCTURE ONLY -- an independent, scalar model of what the reference's
``aggregate_evidence`` and key post-filters compute (seal/keys.py:178-497,
seal/retrieval.py:85-91,178-191), written from the behaviour of that code, not from its
text: every stage is its own function, the index is queried one scalar call at a time the
way the reference does (get_count per key, locate + get_doc_index per row, get_doc per
document), and every place where the reference's RESULT depends on an order of evaluation
(dict insertion order, stable sorts, the order in which overlapping matches are registered,
heap order, float operation order) reproduces that order explicitly and says why.
Runs against an ``OracleFMIndex``.  PINNED against the reference's own code for this layer:
tests/golden/ref_aggregate_evidence.json and ref_helpers.json were produced by running
seal/keys.py itself (tests/golden/make_reference_golden.py) and tests/test_reference_golden.py
checks this file (and the product's host logic) against them bit for bit.  The C++/sdsl layer
underneath stays PARITY UNPINNED (see fm_oracle.c header).
"""
import heapq
import math

NEG_INF = float("-inf")


# --------------------------------------------------------------------------------------
# small helpers of the key pipeline
# --------------------------------------------------------------------------------------
def oracle_strip(seq, symbols_start, symbols_end):
    """keys.py:54-61: drop leading members of ``symbols_start`` and trailing members of
    ``symbols_end`` (the trailing scan never crosses what the leading scan consumed)."""
    lo, hi = 0, len(seq)
    while lo < hi and seq[lo] in symbols_start:
        lo += 1
    while hi > lo and seq[hi - 1] in symbols_end:
        hi -= 1
    return seq[lo:hi]


def _drop_first_if(key, bad):
    return key[1:] if key[0] in bad else key


def _drop_last_if(key, bad):
    return key[:-1] if key[-1] in bad else key


def oracle_body_postfilter(found, index, strip_token_ids=(0, 2), min_length=0):
    """retrieval.py:85-91: up to two special tokens off the front, one off the back (empty
    keys fall out between the passes), optional exact length, then only keys the corpus has."""
    kept = [(s, k) for s, k in found if k]
    for trim in (_drop_first_if, _drop_first_if, _drop_last_if):
        kept = [(s, trim(k, strip_token_ids)) for s, k in kept]
        if trim is not _drop_last_if:
            kept = [(s, k) for s, k in kept if k]
    out = []
    for s, k in kept:
        if not k or (min_length > 0 and len(k) != min_length):
            continue
        if index.get_count(k) > 0:
            out.append((s, k))
    return out


def oracle_title_postfilter(found, index, title_bos=2, title_eos=49314, strip_token_ids=(0, 2), min_length=0):
    """retrieval.py:178-191 with force_decoding_second_token < 0 and partial_titles False:
    complete titles only (they end in the title delimiter once a trailing special is gone),
    optional exact length (delimiter included), title marker in front, corpus membership."""
    out = []
    for s, k in found:
        k = _drop_last_if(k, strip_token_ids)
        if k[-1] != title_eos or (min_length > 0 and len(k) != min_length + 1):
            continue
        if k[0] != title_bos:
            k = [title_bos] + k
        if k and index.get_count(k) > 0:
            out.append((s, k))
    return out


def oracle_deduplicate(list_of_lists):
    """keys.py:19-35: first occurrence of every token sequence wins; items are either bare
    sequences or (float score, sequence) pairs and are returned as they came."""
    seen, kept = set(), []
    for item in list_of_lists:
        seq = item[1] if isinstance(item[0], float) else item
        ident = tuple(seq)
        if ident not in seen:
            seen.add(ident)
            kept.append(item)
    return kept


# --------------------------------------------------------------------------------------
# scoring of keys and unigrams
# --------------------------------------------------------------------------------------
def _log_odds(model_logprob, count, n_tokens, smoothing):
    """keys.py:220-222 / 255-256: log-odds of the model's probability against the smoothed
    corpus frequency, in exactly this association of the four terms."""
    corpus_logprob = math.log((count + smoothing) / (n_tokens + smoothing))
    return (model_logprob + math.log(1 - math.exp(corpus_logprob))) - (corpus_logprob + math.log(1 - math.exp(model_logprob)))


def _relu(x):
    return max(x, 0.0)


class _Scoring:
    def __init__(self, index, use_freq, smoothing, alpha, length_penalty, cutoff):
        self.index, self.use_freq, self.smoothing, self.alpha = index, use_freq, smoothing, alpha
        self.length_penalty, self.cutoff = length_penalty, cutoff
        self.n_tokens = float(index.beginnings[-1])

    def key(self, ngram, model_logprob, count):          # keys.py:212-233
        if count == 0:
            return 0.0
        shrink = (1.0 - self.length_penalty) ** (len(ngram) - 1.0)
        if self.use_freq:
            adjusted = (model_logprob - 1e-10) * shrink
            return _relu(_log_odds(adjusted, count, self.n_tokens, self.smoothing)) ** self.alpha
        return (_relu(model_logprob - self.cutoff) * shrink) ** self.alpha

    def unigram(self, model_logprob, count):             # keys.py:250-264: no exponent on the frequency branch
        if count == 0:
            value = 0.0
        elif self.use_freq:
            value = _relu(_log_odds(model_logprob, count, self.n_tokens, self.smoothing))
        else:
            value = _relu(model_logprob - self.cutoff) ** self.alpha
        return value if value != 0.0 else 0.0           # -0.0 never leaves this function


def _score_unigrams(raw_scores, taken, top_k, scoring, index):
    """keys.py:236-265.  Only the ``top_k`` best tokens by model score (ties: the lower id,
    because the reference sorts ids with a stable descending sort) keep their score, every
    other token is scored from -inf (which the formulas above turn into 0); tokens that are
    already keys, and the three specials, are 0."""
    vocab = len(raw_scores)
    ranking = sorted(range(vocab), key=lambda t: raw_scores[t], reverse=True)
    survivors = set(ranking[:top_k])
    scored = []
    for tok in range(vocab):
        if tok in taken:
            scored.append(0.0)
            continue
        logprob = raw_scores[tok] if tok in survivors else NEG_INF
        scored.append(scoring.unigram(logprob, index.get_count([tok])))
    return scored


def _split_by_frequency(scored_keys, index, rare_limit, frequent_limit):
    """keys.py:280-309.  A key that repeats keeps its FIRST position and its LAST score
    (dict assignment); each class and their union are then ordered by score, descending,
    stable."""
    rare, frequent = {}, {}
    for ngram, score in scored_keys:
        count = index.get_count(ngram)
        if count > frequent_limit or score == 0.0:
            continue
        bucket = frequent if (count > rare_limit or score < 0.0) else rare
        bucket[tuple(ngram)] = score

    def by_score(pairs):
        return dict(sorted(pairs, key=lambda kv: kv[1], reverse=True))
    rare, frequent = by_score(rare.items()), by_score(frequent.items())
    return rare, frequent, by_score(list(rare.items()) + list(frequent.items()))


# --------------------------------------------------------------------------------------
# first stage: occurrences of the rare keys -> per-document evidence
# --------------------------------------------------------------------------------------
def _preference(mode, ngram, score, counts):
    """what "a better single key" means in the three modes (keys.py:322-330, 432-440 use the
    negated form of the same tuples)"""
    if mode == "length":
        return (len(ngram), score)
    if mode == "freq":
        return (-counts[tuple(ngram)], score)
    return score


def _discount(tokens, score, coverage, beta):
    """keys.py:186-191: a key whose token TYPES are already covered counts for less"""
    if not coverage:
        return score
    types = set(tokens)
    fresh = len(types.difference(coverage))
    return (1.0 - beta + (beta * fresh / len(types))) * score


def _first_stage(rare, index, counts, rare_limit, mode, allow_overlaps):
    """keys.py:311-350.  Documents enter the table when one of their rows is LOCATED -- even
    if the occurrence then adds nothing -- so insertion order is the order of first touch.
    An occurrence is "new" when none of its corpus positions was claimed by an earlier
    occurrence (of any key, in any document); a key counts once per document."""
    claimed = set()
    table = {}                                           # doc -> [score, [(key, score), ...], [best key, best score]]
    for ngram, score in rare.items():
        lo, hi = index.get_range(list(ngram))
        counted_for = set()                              # documents this key already contributed to
        for row in range(lo, min(hi, lo + rare_limit)):
            end = index.locate(row)
            span = range(end - len(ngram), end)
            doc = index.get_doc_index(end)
            entry = table.setdefault(doc, [0.0, [], [[], 0.0]])
            is_new = not any(p in claimed for p in span)
            if _preference(mode, ngram, score, counts) > _preference(mode, entry[2][0], entry[2][1], counts):
                entry[2] = [ngram, score]
            if is_new:
                claimed.update(span)
            if (is_new or allow_overlaps) and doc not in counted_for:
                counted_for.add(doc)
                entry[0] += score
                entry[1].append((ngram, score))
    return table


def _apply_type_discounts(table, beta):
    """keys.py:352-364: within a document, in the order the keys arrived, every key is
    discounted by the token types its predecessors covered; the document's score is the sum
    accumulated in that order."""
    for entry in table.values():
        covered, total = set(), 0.0
        for slot, (ngram, score) in enumerate(entry[1]):
            types = set(ngram)
            worth = _discount(types, score, covered, beta)
            total += worth
            entry[1][slot] = [ngram, worth]
            covered |= types
        entry[0] = total


# --------------------------------------------------------------------------------------
# second stage: every key occurrence inside the best documents
# --------------------------------------------------------------------------------------
def _build_trie(all_keys):
    """keys.py:368-375: positive keys only; the score sits under the pseudo-token -1"""
    root = {}
    for ngram, score in all_keys.items():
        if len(ngram) >= 1 and score > 0.0:
            node = root
            for tok in ngram:
                node = node.setdefault(tok, {})
            node[-1] = score
    return root


def _occurrences(doc_tokens, trie):
    """keys.py:395-416.  Returns {key: [score, [(start, end), ...]]} in the reference's
    REGISTRATION order, which decides ties between equally good single keys later on.

    The reference keeps a list of partial matches, oldest first, appends the match that
    starts at the current token, and then rebuilds the list by popping from its END --
    which reverses the survivors.  So the list alternates between oldest-first and
    newest-first from one token to the next, and complete keys ending at a token are
    registered in that alternating order.  The same mechanism, stated as what it is: a
    stack that is emptied into a new stack at every token."""
    registered = {}
    stack = []                                           # [trie node, start index] of the partial matches
    for here, tok in enumerate(doc_tokens):
        advanced = [[node.get(tok) if node is not None else None, start] for node, start in stack]
        advanced.append([trie.get(tok), here])
        survivors = []
        while advanced:                                  # newest entry of `advanced` first
            node, start = advanced.pop()
            if node is None:
                continue
            survivors.append([node, start])
            if -1 in node:
                key = tuple(doc_tokens[start:here + 1])
                registered.setdefault(key, [node[-1], []])[1].append((start, here + 1))
        stack = survivors
    return registered


def _greedy_cover(occurrences, doc_len, beta, allow_overlaps):
    """keys.py:428-471.  Occurrences are taken best score first (ties: smaller key tuple,
    then earlier position -- the natural order of the reference's heap entries
    ``(-score, key, score, start, end)``).  A key's worth is fixed, with the type discount
    against the types covered so far, when its first occurrence is ACCEPTED; the occurrences
    that follow it immediately reuse that worth, and an occurrence that overlaps something
    already placed is skipped without fixing anything."""
    heap = []
    for key, (score, spans) in occurrences.items():
        for start, end in spans:
            heapq.heappush(heap, (-score, key, score, start, end))
    covered, chosen, last_key = set(), [], None
    free = [True] * doc_len
    while heap:
        _, key, score, start, end = heapq.heappop(heap)
        types = set(key)
        if last_key == key:
            worth = chosen[-1][1]
        elif not types:
            worth = 0.0
        else:
            worth = _discount(types, score, covered, beta)
        if worth <= 0.0:
            continue
        if not allow_overlaps and not all(free[start:end]):
            continue
        if last_key == key:
            chosen[-1] = (key, worth)
        else:
            last_key = key
            covered |= types
            chosen.append((key, worth))
        free[start:end] = [False] * (end - start)
    return chosen, covered, free


def _score_document(doc_tokens, trie, unigram_scores, counts, mode, beta, allow_overlaps, single_key,
                    single_key_add_unigrams, unigrams_ignore_free_places):
    """keys.py:386-494 for one document: [score, keys used, None, tokens, best single key]"""
    occurrences = _occurrences(doc_tokens, trie)
    best = [[], 0.0]
    for key, (score, _) in occurrences.items():          # registration order; strictly better replaces
        if _preference(mode, key, score, counts) > _preference(mode, best[0], best[1], counts):
            best = [key, score]
    chosen, covered, free = _greedy_cover(occurrences, len(doc_tokens), beta, allow_overlaps)
    if unigrams_ignore_free_places:
        free = [True] * len(free)
    multi = sum([worth for _, worth in chosen])
    # keys.py:473-486: token types on still-free positions, in order of first free occurrence
    fill = 0.0
    seen_types = set()
    for tok, is_free in zip(doc_tokens, free):
        if not is_free or tok in seen_types:
            continue
        seen_types.add(tok)
        base = unigram_scores[tok] if unigram_scores is not None else 0.0
        if base > 0.0:
            worth = _discount((tok,), base, covered, beta)
            if worth != 0.0:
                fill += worth
                chosen.append(((tok,), worth))
    single = best[1] + (fill if single_key_add_unigrams else 0.0)
    multi += fill
    return [(1.0 - single_key) * multi + single_key * single, chosen, None, doc_tokens, best]


# --------------------------------------------------------------------------------------
# the whole thing
# --------------------------------------------------------------------------------------
def oracle_aggregate_evidence(ngrams_and_scores, unigram_scores=None, index=None, max_occurrences_1=1500,
                              max_occurrences_2=10_000_000, n_docs_complete_score=500, alpha=2.0, beta=0.8,
                              length_penalty=0.0, use_fm_index_frequency=True, add_best_unigrams_to_ngrams=False,
                              use_top_k_unigrams=1000, sort_by_length=False, sort_by_freq=False, smoothing=5.0,
                              allow_overlaps=False, single_key=0.0, single_key_add_unigrams=False,
                              unigrams_ignore_free_places=False, first_stage_only=False):
    mode = "length" if sort_by_length else ("freq" if sort_by_freq else "score")
    keys = [(list(ng), lp) for ng, lp in ngrams_and_scores]
    cutoff = None
    if not use_fm_index_frequency:                       # keys.py:203-205: an empty key list is an IndexError there
        cutoff = sorted(lp for _, lp in keys)[0] - 0.1
    scoring = _Scoring(index, use_fm_index_frequency, smoothing, alpha, length_penalty, cutoff)

    counts = {(): len(index)}
    single_token_keys = {0, 1, 2}
    scored_keys = []
    for ngram, logprob in keys:                          # keys.py:207-234
        if len(ngram) == 1:
            single_token_keys.add(ngram[0])
        count = counts[tuple(ngram)] = index.get_count(ngram)
        scored_keys.append((ngram, scoring.key(ngram, logprob, count)))

    if unigram_scores is not None:
        unigram_scores = _score_unigrams(list(unigram_scores), single_token_keys, use_top_k_unigrams, scoring, index)
        if add_best_unigrams_to_ngrams:                  # keys.py:266-278: as many unigrams as there are keys, best first, ties to the lower id
            ranking = sorted(range(len(unigram_scores)), key=lambda t: -unigram_scores[t])
            for tok in ranking[:len(scored_keys)]:
                counts[(tok,)] = index.get_count([tok])
                scored_keys.append(([tok], unigram_scores[tok]))

    rare, _, all_keys = _split_by_frequency(scored_keys, index, max_occurrences_1, max_occurrences_2)

    table = _first_stage(rare, index, counts, max_occurrences_1, mode, allow_overlaps)
    _apply_type_discounts(table, beta)

    def rank_value(item):                                # keys.py:366: blend of evidence and best single key, ascending
        entry = item[1]
        return (1.0 - single_key) * (-entry[0]) + single_key * (-entry[2][1])
    shortlist = sorted(table.items(), key=rank_value)[:n_docs_complete_score]
    if first_stage_only:
        return dict(shortlist), all_keys

    trie = _build_trie(all_keys)
    results = {}
    for doc, _ in shortlist:
        doc_tokens = [2] + index.get_doc(doc)[:-1]       # keys.py:388: the text as decoded: leading </s>, own eos dropped
        results[doc] = _score_document(doc_tokens, trie, unigram_scores, counts, mode, beta, allow_overlaps, single_key,
                                       single_key_add_unigrams, unigrams_ignore_free_places)
    return dict(sorted(results.items(), key=lambda kv: -kv[1][0])), all_keys
