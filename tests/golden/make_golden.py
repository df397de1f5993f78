#!/usr/bin/env python
"""Generates tests/golden/*.json.

The reference (facebookresearch/SEAL) ships no golden vectors and cannot be run
offline (sdsl-lite / SWIG absent), so these fixtures are produced by BRUTE FORCE
-- naive suffix sort and naive substring counting in pure Python, independent of
both oracle/ and seal_amd/ -- through the semantics of reference seal/index.py
(per-document reversal, SHIFT=10, 0 sentinel).  They pin layout-independent
facts: suffix array, BWT, occurrence counts, matching rows, document ids.

  python tests/golden/make_golden.py        # rewrites the json files next to it
"""
import json
import os
import random

SHIFT = 10
HERE = os.path.dirname(os.path.abspath(__file__))


def build(docs):
    text, beg = [], [0]
    for d in docs:
        beg.append(beg[-1] + len(d))
        text.extend(t + SHIFT for t in reversed(d))
    text.append(0)
    sa = sorted(range(len(text)), key=lambda i: text[i:])
    bwt = [text[i - 1] for i in sa]
    return text, beg, sa, bwt


def rows_of(text, sa, pattern):
    pat = [t + SHIFT for t in reversed(pattern)]
    rows = [r for r, i in enumerate(sa) if text[i:i + len(pat)] == pat]
    return (rows[0], rows[-1] + 1) if rows else None


def case(name, docs, patterns, note):
    text, beg, sa, bwt = build(docs)
    import bisect
    pats = []
    for p in patterns:
        r = rows_of(text, sa, p)
        entry = {"pattern": p, "count": 0 if r is None else r[1] - r[0]}
        if r is not None and len(p) >= 2:     # single-token ranges may be widened by quirk Q1 in the reference
            lo, hi = r
            entry["range"] = [lo, hi]
            entry["positions"] = [sa[x] for x in range(lo, hi)]
            entry["docs"] = [bisect.bisect_right(beg, sa[x]) - 1 for x in range(lo, hi)]
            seg = bwt[lo:hi]
            syms = sorted(set(seg))
            entry["continuations"] = [[s - SHIFT, seg.count(s)] for s in syms if s > 0]
        pats.append(entry)
    return {"name": name, "note": note, "docs": docs, "beginnings": beg, "suffix_array": sa, "bwt": bwt, "patterns": pats}


def main():
    out = []
    out.append(case("survey_g2", [[5, 6, 7, 2], [5, 6, 8, 2], [9, 5, 6, 2]],
                    [[5, 6], [2, 5, 6], [2, 9], [6, 7], [5], [7, 2], [8, 8]],
                    "SURVEY.md section 8c, vector G2 (title-bos trick Q7: [2,9] unmatchable for the last doc)"))
    # res/sample/sample_corpus.tsv through the README.md:149-161 recipe ("Title @@ text" + eos) with a stand-in
    # word-level id table (the BART tokenizer files are not available offline); '@@' keeps its real id 49314
    words = {}

    def wid(w):
        return words.setdefault(w, 100 + 7 * len(words))
    sample = ["Doc 1 @@ This is a sample document", "Doc 2 @@ This is another sample document",
              "Doc 3 @@ And here you find the final one"]
    docs = [[49314 if w == "@@" else wid(w) for w in line.split()] + [2] for line in sample]
    out.append(case("res_sample_corpus", docs,
                    [[wid("This"), wid("is")], [wid("sample"), wid("document")], [wid("is"), wid("another")],
                     [2, wid("Doc")], [wid("Doc"), wid("2"), 49314], [wid("final"), wid("one"), 2], [wid("one"), wid("one")]],
                    "reference res/sample/sample_corpus.tsv (config[0]), stand-in word ids: " + json.dumps(words)))
    rng = random.Random(1234)
    docs = [[rng.randrange(3, 40) for _ in range(rng.randrange(2, 15))] + [2] for _ in range(40)]
    pats = []
    for _ in range(40):
        d = rng.choice(docs)
        a = rng.randrange(len(d) - 1)
        pats.append(d[a:a + rng.randrange(2, 5)])
    pats += [[39, 39, 39, 39], [3, 4]]
    out.append(case("random_small_alphabet", docs, pats, "seeded random corpus, alphabet 37: many repeats"))
    rng = random.Random(99)
    docs = [[rng.choice([4, 5, 6, 7, 50000, 50264, 49314, 30000 + rng.randrange(100)]) for _ in range(rng.randrange(3, 12))] + [2]
            for _ in range(30)]
    pats = [d[a:a + 2] for d in docs[:25] for a in (0, 1)]
    out.append(case("bart_sized_alphabet", docs, pats, "symbols up to 50274: 16 wavelet levels"))
    with open(os.path.join(HERE, "fm_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
