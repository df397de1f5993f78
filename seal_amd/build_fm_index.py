"""``python -m seal_amd.build_fm_index corpus.tsv out_prefix`` -- corpus -> FM-index (reference scripts/build_fm_index.py).

Same options.  Passages are ``id<TAB>title<TAB>text`` lines (``--format kilt``) or DPR's ``id<TAB>text<TAB>title`` CSV with a
header (``--format dpr``); every passage becomes ``[title @@] text`` token ids + eos; the index is built on the GPU
(``FMIndex.initialize``; suffix array, BWT and wavelet matrix by ``fmi_build_gpu.hip``) and saved as
``out_prefix.fmi`` + ``out_prefix.oth`` with the passage ids as labels.  ``ftfy`` / spaCy / the fairseq hub model are used
when the corresponding option asks for them and they are installed; the tokenizer is ``--hf_model`` (e.g.
``facebook/bart-large``)."""
import argparse
import csv
import re
from typing import Callable, Iterator, List, Optional

from .index import FMIndex


def preprocess_file(input_path: str, labels: List[str], format: str = "kilt", lowercase: bool = False,
                    word_tokenize: Optional[Callable[[str], List[str]]] = None, include_title: bool = False, delim: str = "@@") -> Iterator[str]:
    """the passage strings to tokenize, in file order; their ids are appended to ``labels``
    (scripts/build_fm_index.py:28-73: whitespace squeezed, wiki markers removed, empty texts dropped)"""
    try:
        from ftfy import fix_text
    except ImportError:                               # the reference repairs mojibake with ftfy; without it the text is taken as it is
        fix_text = None
    with open(input_path, "r", 2 ** 16, newline="" if format == "dpr" else None) as f:
        if format == "dpr":
            next(f)
            pieces = ((pp[0], pp[2], pp[1]) for pp in csv.reader(f, delimiter="\t", quotechar='"') if len(pp) == 3)
        elif format == "kilt":
            pieces = (pp for pp in (line.strip().split("\t", 2) for line in f) if len(pp) == 3)
        else:
            raise ValueError(f"unknown corpus format {format!r}")
        for idx, title, text in pieces:
            idx, title = idx.strip(), title.strip()
            text = re.sub(r"\s+", " ", text)
            if fix_text is not None:
                text = fix_text(text)
            text = text.replace("BULLET::::", "").replace("SECTION::::", "").strip()
            if not text:
                continue
            if word_tokenize is not None:
                title, text = " ".join(word_tokenize(title)), " ".join(word_tokenize(text))
            title = f"{title} {delim}"
            if include_title and title:
                text = f"{title} {text}"
            if lowercase:
                text = text.lower()
            labels.append(idx)
            yield text


_WORKER_TOKENIZE: Optional[Callable[[str], List[int]]] = None


def _tokenize_in_worker(line: str) -> List[int]:
    """module-level (picklable by name) entry of the tokenisation workers: the tokenizer itself -- a closure over a HF
    tokenizer or a hub model -- is inherited through ``fork`` in ``_WORKER_TOKENIZE``, never pickled"""
    return _WORKER_TOKENIZE(line)


def build_index(input_path: str, tokenize: Callable[[str], List[int]], format: str = "kilt", lowercase: bool = False,
                word_tokenize=None, include_title: bool = False, delim: str = "@@", jobs: int = 1) -> FMIndex:
    global _WORKER_TOKENIZE
    labels: List[str] = []
    lines = preprocess_file(input_path, labels, format, lowercase, word_tokenize, include_title, delim)
    index = FMIndex()
    if jobs > 1:
        import multiprocessing
        _WORKER_TOKENIZE = tokenize            # set BEFORE the fork: the children see it, nothing is pickled but the lines
        try:
            with multiprocessing.get_context("fork").Pool(jobs) as pool:     # tokenisation only: forked before any HIP context exists
                sequences = pool.imap(_tokenize_in_worker, lines, chunksize=256)
                index.initialize(sequences)
        finally:
            _WORKER_TOKENIZE = None
    else:
        index.initialize(tokenize(line) for line in lines)
    index.labels = labels
    return index


def make_tokenizer(hf_model: Optional[str]) -> Callable[[str], List[int]]:
    if hf_model is not None:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(hf_model, use_fast=False)
        is_bart = "bart" in hf_model

        def tokenize(text: str) -> List[int]:
            text = text.strip()
            if is_bart:
                text = " " + text
            return tok(text, add_special_tokens=False)["input_ids"] + [tok.eos_token_id]
        return tokenize
    import torch
    bart = torch.hub.load("pytorch/fairseq", "bart.large").eval()     # the reference's default when no --hf_model is given
    return lambda text: bart.encode(" " + text.strip()).tolist()[1:]


def parse_args(argv=None):
    p = argparse.ArgumentParser(prog="seal_amd.build_fm_index")
    p.add_argument("input")
    p.add_argument("output")
    p.add_argument("--jobs", type=int, default=1)
    p.add_argument("--include_title", action="store_true")
    p.add_argument("--delim", default="@@")
    p.add_argument("--format", choices=["kilt", "dpr"], default="kilt")
    p.add_argument("--hf_model", default=None, type=str)
    p.add_argument("--lowercase", action="store_true")
    p.add_argument("--tokenize", action="store_true", help="word-tokenize title and text with spaCy first")
    return p.parse_args(argv)


def main(argv=None) -> None:
    args = parse_args(argv)
    print(args)
    word_tokenize = None
    if args.tokenize:
        from spacy.lang.en import English
        nlp_tok = English().tokenizer
        word_tokenize = lambda text: [t.text.strip() for t in nlp_tok(text)]      # noqa: E731
    index = build_index(args.input, make_tokenizer(args.hf_model), args.format, args.lowercase, word_tokenize, args.include_title,
                        args.delim, args.jobs)
    index.save(args.output)


if __name__ == "__main__":
    main()
