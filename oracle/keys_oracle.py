"""TEST INFRASTRUCTURE ONLY -- restatement of ``aggregate_evidence`` and the key
post-filters (reference seal/keys.py:178-497, seal/retrieval.py:85-91,178-191)
with per-call scalar index queries, exactly as the reference issues them
(get_count per key, locate + get_doc_index per row, get_doc per document).
Runs against an ``OracleFMIndex``.  PARITY UNPINNED (see fm_oracle.c header).
"""
import math
from collections import Counter, defaultdict
from heapq import heappop, heappush
from itertools import chain, islice


def oracle_strip(seq, symbols_start, symbols_end):      # keys.py:54-61
    i = 0
    while i < len(seq) and seq[i] in symbols_start:
        i += 1
    j = len(seq)
    while j > i and seq[j - 1] in symbols_end:
        j -= 1
    return seq[i:j]


def oracle_body_postfilter(found, index, strip_token_ids=(0, 2), min_length=0):   # retrieval.py:85-91
    fk = list(found)
    fk = [(s, k[1:] if k[0] in strip_token_ids else k) for s, k in fk if k]
    fk = [(s, k[1:] if k[0] in strip_token_ids else k) for s, k in fk if k]
    fk = [(s, k[:-1] if k[-1] in strip_token_ids else k) for s, k in fk if k]
    if min_length > 0:
        fk = [(s, k) for s, k in fk if len(k) == min_length]
    return [(s, k) for s, k in fk if k and index.get_count(k) > 0]


def oracle_title_postfilter(found, index, title_bos=2, title_eos=49314, strip_token_ids=(0, 2), min_length=0):
    # retrieval.py:178-191 with force_decoding_second_token < 0 and partial_titles False
    fk = [(s, k[:-1] if k[-1] in strip_token_ids else k) for s, k in found]
    fk = [(s, k) for s, k in fk if k[-1] == title_eos]
    if min_length > 0:
        fk = [(s, k) for s, k in fk if len(k) == (min_length + 1)]
    fk = [(s, [title_bos] + k if k[0] != title_bos else k) for s, k in fk]
    return [(s, k) for s, k in fk if k and index.get_count(k) > 0]


def oracle_deduplicate(list_of_lists):                  # keys.py:19-35
    present, result = set(), []
    for el in list_of_lists:
        x = el
        if isinstance(el[0], float):
            el = el[1]
        t = tuple(el)
        if t in present:
            continue
        present.add(t)
        result.append(x)
    return result


def oracle_aggregate_evidence(ngrams_and_scores, unigram_scores=None, index=None, max_occurrences_1=1500,
                              max_occurrences_2=10_000_000, n_docs_complete_score=500, alpha=2.0, beta=0.8,
                              length_penalty=0.0, use_fm_index_frequency=True, add_best_unigrams_to_ngrams=False,
                              use_top_k_unigrams=1000, sort_by_length=False, sort_by_freq=False, smoothing=5.0,
                              allow_overlaps=False, single_key=0.0, single_key_add_unigrams=False,
                              unigrams_ignore_free_places=False, first_stage_only=False):
    def repetition(ngram, score, coverage):             # keys.py:186-191
        if not coverage:
            return score
        ngram = set(ngram)
        coeff = 1.0 - beta + (beta * len(ngram.difference(coverage)) / len(ngram))
        return coeff * score

    ntokens = float(index.beginnings[-1])
    ngrams_and_scores = [(list(ng), sr) for ng, sr in ngrams_and_scores]
    counts = {tuple(): len(index)}
    cutoff = None
    if not use_fm_index_frequency:
        cutoff = sorted(ngrams_and_scores, key=lambda x: x[1])[0][1] - 0.1
    unigrams = {0, 1, 2}
    for i in range(len(ngrams_and_scores)):             # keys.py:207-234
        ngram, sr = ngrams_and_scores[i]
        if len(ngram) == 1:
            unigrams.add(ngram[0])
        count = index.get_count(ngram)
        counts[tuple(ngram)] = count
        if count == 0:
            sco = 0.0
        elif use_fm_index_frequency:
            sr -= 1e-10
            sr *= (1.0 - length_penalty) ** (len(ngram) - 1.0)
            snr = math.log((count + smoothing) / (ntokens + smoothing))
            sco = (sr + math.log(1 - math.exp(snr))) - (snr + math.log(1 - math.exp(sr)))
            sco = max(sco, 0.0)
            sco **= alpha
        else:
            sco = sr - cutoff
            sco = max(sco, 0.0)
            sco *= (1.0 - length_penalty) ** (len(ngram) - 1.0)
            sco **= alpha
        ngrams_and_scores[i] = (ngram, sco)

    if unigram_scores is not None:                      # keys.py:236-278
        unigram_scores = unigram_scores[:]
        best = sorted(range(len(unigram_scores)), reverse=True, key=lambda i: unigram_scores[i])
        best = set(best[:use_top_k_unigrams])
        unigram_scores = [s if i in best else float('-inf') for i, s in enumerate(unigram_scores)]
        for i in range(len(unigram_scores)):
            if i in unigrams:
                unigram_scores[i] = 0.0
                continue
            sr = unigram_scores[i]
            count = index.get_count([i])
            if count == 0:
                sco = 0.0
            elif use_fm_index_frequency:
                snr = math.log((count + smoothing) / (ntokens + smoothing))
                sco = (sr + math.log(1 - math.exp(snr))) - (snr + math.log(1 - math.exp(sr)))
                sco = max(sco, 0.0)
            else:
                sco = sr - cutoff
                sco = max(sco, 0.0)
                sco **= alpha
            unigram_scores[i] = sco if sco != 0.0 else 0.0
        if add_best_unigrams_to_ngrams:
            best_unigrams = sorted(list(range(len(unigram_scores))), key=lambda x: -unigram_scores[x])[:len(ngrams_and_scores)]
            for i in best_unigrams:
                counts[tuple([i])] = index.get_count([i])
                ngrams_and_scores.append(([i], unigram_scores[i]))

    rare_ngrams = defaultdict(float)                    # keys.py:280-309
    freq_ngrams = defaultdict(float)
    for ngram, sco in ngrams_and_scores:
        count = index.get_count(ngram)
        if count > max_occurrences_2:
            continue
        elif sco == 0.0:
            continue
        elif count > max_occurrences_1 or sco < 0.0:
            ngrams = freq_ngrams
        else:
            ngrams = rare_ngrams
        ngrams[tuple(ngram)] = sco
    rare_ngrams = {k: v for k, v in sorted(rare_ngrams.items(), key=lambda x: x[1], reverse=True)}
    freq_ngrams = {k: v for k, v in sorted(freq_ngrams.items(), key=lambda x: x[1], reverse=True)}
    all_ngrams = {k: v for k, v in sorted(chain(rare_ngrams.items(), freq_ngrams.items()), key=lambda x: x[1], reverse=True)}

    covered_points = set()                              # keys.py:311-350
    first_stage = defaultdict(lambda: [0.0, [], [[], 0.0]])
    for ngram, sco in rare_ngrams.items():
        doc_done = defaultdict(set)
        for row in islice(range(*index.get_range(list(ngram))), max_occurrences_1):
            tok_end = index.locate(row)
            tok_start = tok_end - len(ngram)
            doc = index.get_doc_index(tok_end)
            new = all([i not in covered_points for i in range(tok_start, tok_end)])
            if sort_by_length:
                order = (len(ngram), sco)
                max_order = (len(first_stage[doc][2][0]), first_stage[doc][2][1])
            elif sort_by_freq:
                order = (-counts[tuple(ngram)], sco)
                max_order = (-counts[tuple(first_stage[doc][2][0])], first_stage[doc][2][1])
            else:
                order = sco
                max_order = first_stage[doc][2][1]
            if order > max_order:
                first_stage[doc][2] = [ngram, sco]
            if new:
                for tok in range(tok_start, tok_end):
                    covered_points.add(tok)
            if new or allow_overlaps:
                if ngram not in doc_done[doc]:
                    doc_done[doc].add(ngram)
                    first_stage[doc][0] += sco
                    first_stage[doc][1].append((ngram, sco))

    for doc, doc_info in first_stage.items():           # keys.py:352-364
        current_coverage = set()
        current_score = 0.0
        for i in range(len(doc_info[1])):
            tt, sco = doc_info[1][i]
            tts = set(tt)
            new_sco = repetition(tts, sco, current_coverage)
            current_score += new_sco
            doc_info[1][i] = [tt, new_sco]
            current_coverage |= tts
        doc_info[0] = current_score

    to_fully_score = sorted(first_stage.items(),
                            key=lambda x: (1.0 - single_key) * (-x[1][0]) + single_key * (-x[1][2][1]))[:n_docs_complete_score]
    if first_stage_only:
        return {doc: info for doc, info in to_fully_score}, all_ngrams

    results = defaultdict(lambda: [0.0, [], None, None, [[], 0.0]])   # keys.py:368-375
    trie = {}
    for ngram, score in all_ngrams.items():
        if len(ngram) < 1 or score <= 0.0:
            continue
        current = trie
        for t in ngram:
            current = current.setdefault(t, {})
        current[-1] = score

    for doc, _ in to_fully_score:                       # keys.py:386-494
        doc_tokens = [2] + index.get_doc(doc)[:-1]
        results[doc][3] = doc_tokens
        if unigram_scores is not None:
            type_scores = {t: unigram_scores[t] for t in doc_tokens}
        else:
            type_scores = {t: 0.0 for t in doc_tokens}
        matches = {}
        open_matches = []
        for i in range(len(doc_tokens)):
            open_matches = [(m.get(doc_tokens[i]), l + 1, n) for (m, l, n) in open_matches] + [(trie.get(doc_tokens[i]), 1, [])]
            for _, _, n in open_matches:
                n.append(doc_tokens[i])
            new_open_matches = []
            while open_matches:
                m, l, n = open_matches.pop()
                if m is None:
                    continue
                new_open_matches.append((m, l, n))
                if -1 in m:
                    matches.setdefault(tuple(n), [m[-1], []])[1].append((i - l + 1, i + 1))
            open_matches = new_open_matches
        greedy_matches = []
        for n, (s, d) in matches.items():
            if sort_by_length:
                order = (-len(n), -s)
                max_order = (-len(results[doc][4][0]), -results[doc][4][1])
            elif sort_by_freq:
                order = (counts[tuple(n)], -s)
                max_order = (counts[tuple(results[doc][4][0])], -results[doc][4][1])
            else:
                order = -s
                max_order = -results[doc][4][1]
            for (i, j) in d:
                heappush(greedy_matches, (-s, n, s, i, j))
            if order < max_order:
                results[doc][4] = [n, s]
        current_coverage = set()
        ngrams = []
        prev = None
        free = [True] * len(doc_tokens)
        while greedy_matches:
            order, n, s, i, j = heappop(greedy_matches)
            n_set = set(n)
            if prev == n:
                new_s = ngrams[-1][1]
            elif not n_set:
                new_s = 0.0
            else:
                new_s = repetition(n_set, s, current_coverage)
            if new_s <= 0.0:
                continue
            if not (allow_overlaps or all(free[i:j])):
                continue
            if prev == n:
                ngrams[-1] = (n, new_s)
            else:
                prev = n
                current_coverage |= n_set
                ngrams.append((n, new_s))
            free[i:j] = [False] * (j - i)
        if unigrams_ignore_free_places:
            free = [True for _ in free]
        single_key_score = results[doc][4][1]
        multi_key_score = sum([s for n, s in ngrams])
        unigram_score = 0.0
        for t, f in Counter([t for t, b in zip(doc_tokens, free) if b]).items():
            s = type_scores[t]
            if s > 0.0:
                n = (t,)
                s = repetition(n, s, current_coverage)
                if s != 0.0:
                    unigram_score += s
                    ngrams.append((n, s))
        if single_key_add_unigrams:
            single_key_score += unigram_score
        multi_key_score += unigram_score
        results[doc][0] = (1.0 - single_key) * multi_key_score + single_key * single_key_score
        results[doc][1] = ngrams
    results = {k: v for k, v in sorted(results.items(), key=lambda x: -x[1][0])}
    return results, all_ngrams
