"""Query-derived keys (``add_query_to_keys``; reference seal/keys.py:38-51 and
seal/retrieval.py:113-131): every word 1..3-gram of the query, in every
capitalisation pattern, tokenised as a decoder-side string.  Needs spaCy's
English tokenizer and the BART tokenizer, exactly as the reference does."""
from itertools import product

_word_tokenizer = None


def _words(query: str):
    global _word_tokenizer
    if _word_tokenizer is None:
        from spacy.lang.en import English   # reference retrieval.py:39-43
        _word_tokenizer = English().tokenizer
    return [t.text for t in _word_tokenizer(query.strip())]


def decompose_query_into_keys(query: str, length: int = 3):
    words = _words(query)
    out = set()
    for i in range(len(words)):
        for j in range(i + 1, min(len(words), i + length) + 1):
            span = words[i:j]
            for caps in product((True, False), repeat=j - i):
                out.add(" " + " ".join(w[0].upper() + w[1:] if c else w for c, w in zip(caps, span)))
    return list(out)


def query_ngram_keys(query: str, searcher):
    tok = searcher.bart_tokenizer
    ids = tok(decompose_query_into_keys(query, 3), padding=False, add_special_tokens=False)["input_ids"]
    with tok.as_target_tokenizer():
        ids = tok(tok.batch_decode(ids), padding=False)["input_ids"]
    strip = searcher.strip_token_ids
    ids = [k[:-1] if k and k[-1] in strip else k for k in ids if k]
    ids = [k[1:] if k and k[0] in strip else k for k in ids if k]
    ids = [k[1:] if k and k[0] in strip else k for k in ids if k]
    if searcher.min_length > 0:
        ids = [k for k in ids if len(k) == searcher.min_length]
    return ids


def token_ngram_keys(tokens, searcher=None, length: int = 3):
    """query keys for a PRE-TOKENISED query (an extension: the reference only takes strings): every contiguous span of
    1..``length`` content tokens of the query (its ids without the leading <s> and the trailing </s>), first occurrence
    first -- what reference keys.py:38-51 + retrieval.py:115-126 yield when every word is one token and capitalisation
    does not change it.  The count > 0 filter and the rescoring follow in the caller as for string queries."""
    strip = searcher.strip_token_ids if searcher is not None else (0, 2)
    body = list(tokens)
    while body and body[0] in strip:
        body = body[1:]
    while body and body[-1] in strip:
        body = body[:-1]
    seen, out = set(), []
    for i in range(len(body)):
        for j in range(i + 1, min(len(body), i + length) + 1):
            k = tuple(body[i:j])
            if k not in seen:
                seen.add(k)
                out.append(list(k))
    if searcher is not None and searcher.min_length > 0:
        out = [k for k in out if len(k) == searcher.min_length]
    return out
