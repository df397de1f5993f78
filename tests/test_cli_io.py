"""The search CLI's file formats (reference seal/data.py, seal/search.py): every topics reader on a file written in its
format, every run writer on fake hits, and the CLI loop end to end over a stub searcher (no model, no index)."""
import json

import pytest

from seal_amd.data import OutputFormat, TopicsFormat, get_output_writer, get_query_iterator


class Hit:
    def __init__(self, docid, score, title="T", body="B", keys=None, query=None):
        self.docid, self.score, self._t, self.keys, self.query = docid, score, (title, body), keys, query

    def text(self):
        return self._t


def test_topic_readers(tmp_path):
    p = tmp_path / "q.tsv"
    p.write_text("9\twhen \"was\" it\n7\twho wrote it\n10\tlast\n")
    # pyserini's default iterator: integer ids from a .tsv file, walked in sorted order (not file order, not "10" < "7")
    assert list(get_query_iterator(str(p), TopicsFormat.DEFAULT)) == [(7, "who wrote it"), (9, 'when "was" it'), (10, "last")]
    p.write_text("q9\tnine\nq10\tten\n")
    assert list(get_query_iterator(str(p), TopicsFormat.DEFAULT)) == [("q10", "ten"), ("q9", "nine")]
    p = tmp_path / "q.json"
    p.write_text(json.dumps({"b": "z", "a": {"title": "x y"}}))
    assert list(get_query_iterator(str(p), TopicsFormat("default"))) == [("a", "x y"), ("b", "z")]
    p = tmp_path / "kilt.jsonl"
    p.write_text(json.dumps({"id": "k1", "input": "q one", "meta": {"template_questions": ["tq one"]}}) + "\n\n" +
                 json.dumps({"id": "k2", "input": "q two", "meta": {"template_questions": ["tq two", "other"]}}) + "\n")
    assert list(get_query_iterator(str(p), TopicsFormat.KILT)) == [("k1", "q one"), ("k2", "q two")]
    assert list(get_query_iterator(str(p), TopicsFormat.KILT_TEMPLATE)) == [("k1", "tq one"), ("k2", "tq two")]
    p = tmp_path / "dpr.json"
    p.write_text(json.dumps([{"question": "q0", "answers": ["a"]}, {"question": "q1", "answers": []}]))
    it = get_query_iterator(str(p), TopicsFormat.DPR)
    assert list(it) == [(0, "q0"), (1, "q1")] and it.topics[0]["answers"] == ["a"] and len(it) == 2
    p = tmp_path / "qas.tsv"
    p.write_text('who?\t["x", "y z"]\n"a ""quoted"" one"\t[\'w\']\n')
    it = get_query_iterator(str(p), TopicsFormat.DPR_QAS)
    assert list(it) == [(0, "who?"), (1, 'a "quoted" one')] and it.topics[0]["answers"] == ["x", "y z"]
    (tmp_path / "bad.tsv").write_text("q\tnot a list\n")
    with pytest.raises(Exception):
        get_query_iterator(str(tmp_path / "bad.tsv"), TopicsFormat.DPR_QAS)
    p = tmp_path / "nq.jsonl"
    p.write_text(json.dumps({"example_id": 11, "question_text": "n one"}) + "\n" + json.dumps({"example_id": 5, "question_text": "n two"}) + "\n")
    assert list(get_query_iterator(str(p), TopicsFormat.NQ)) == [(11, "n one"), (5, "n two")]


def test_run_writers(tmp_path):
    hits = [Hit("12-3", 2.5, "Ti ", " body a", keys=[["k", 1.0]], query="the q"), Hit("12-4-6", 1.25, "Ti", "body b"), Hit("40", -0.5, "U", "c")]
    out = tmp_path / "run.trec"
    with get_output_writer(str(out), OutputFormat.TREC, "w", max_hits=2, tag="SEAL") as w:
        w.write("t1", hits)
        w.write("t2", hits[2:])
    assert out.read_text() == "t1 Q0 12-3 1 2.500000 SEAL\nt1 Q0 12-4-6 2 1.250000 SEAL\nt2 Q0 40 1 -0.500000 SEAL\n"
    out = tmp_path / "run.msmarco"
    with get_output_writer(str(out), OutputFormat("msmarco"), "w", max_hits=10) as w:
        w.write(3, hits)
    assert out.read_text() == "3\t12-3\t1\n3\t12-4-6\t2\n3\t40\t3\n"
    out = tmp_path / "run.maxp"          # best passage per document: '12-3' and '12-4-6' are passages of document '12'
    with get_output_writer(str(out), OutputFormat.TREC, "w", max_hits=10, tag="SEAL", use_max_passage=True, max_passage_delimiter="-",
                           max_passage_hits=5) as w:
        w.write("t", hits)
    assert out.read_text() == "t Q0 12 1 2.500000 SEAL\nt Q0 40 2 -0.500000 SEAL\n"
    out = tmp_path / "run.kilt"
    with get_output_writer(str(out), OutputFormat.KILT, "w", max_hits=10) as w:
        w.write("k1", hits)
    rec = json.loads(out.read_text())
    assert rec["id"] == "k1" and rec["input"] == "the q"
    prov = rec["output"][0]["provenance"]
    assert [(p["wikipedia_id"], p["start_paragraph_id"], p["end_paragraph_id"]) for p in prov] == [(12, 3, 3), (12, 4, 6), (40, 0, 0)]
    assert prov[0]["text"] == "Ti  @@  body a" and prov[0]["meta"] == {"keys": [["k", 1.0]]} and "meta" not in prov[1] and prov[1]["score"] == 1.25
    topics = {0: {"question": "q0"}, 1: {"question": "q1"}}
    out = tmp_path / "run.dpr"
    with get_output_writer(str(out), OutputFormat.DPR, "w", max_hits=1, topics=topics) as w:
        w.write(1, hits)
        w.write(0, hits[1:])
    data = json.loads(out.read_text())
    assert [d["question"] for d in data] == ["q1", "q0"]
    assert data[0]["ctxs"] == [{"title": "Ti", "text": "body a", "score": 2.5, "passage_id": "12-3"}] and len(data[1]["ctxs"]) == 1


@pytest.mark.parametrize("chunked", [0, 2])
def test_cli_loop_over_a_stub_searcher(tmp_path, chunked):
    from seal_amd import search

    class Stub:
        calls = []

        def batch_search(self, texts, k=100):
            self.calls.append(list(texts))
            return [[Hit(f"{len(t)}-0", 1.0 / (1 + i)) for i in range(3)] for t in texts]
    topics = tmp_path / "q.tsv"
    topics.write_text("".join(f"{i}\tquery number {'x' * i}\n" for i in range(5)))
    out = tmp_path / "run.trec"
    args = search.build_parser().parse_args(["--topics", str(topics), "--output", str(out), "--hits", "2", "--chunked", str(chunked),
                                             "--fm_index", "unused", "--keep_samples", "4"])
    assert args.beam == 15 and args.length == 10 and args.add_query_to_keys is True          # SEALSearcher's options are all there
    stub = Stub()
    Stub.calls = []
    assert search.run(args, searcher=stub) == 4
    lines = out.read_text().splitlines()
    assert len(lines) == 8 and all(line.split()[1] == "Q0" and line.endswith("SEAL") for line in lines)
    assert [len(c) for c in Stub.calls] == ([4] if chunked == 0 else [2, 2])
    assert sorted({line.split()[0] for line in lines}) == sorted(str(t) for t in args_topics(args))


def args_topics(args):
    import random
    order = [str(i) for i in range(5)]
    random.seed(42)
    random.shuffle(order)
    return order[:args.keep_samples]


def test_corpus_preprocessing_of_the_index_builder(tmp_path):
    """seal_amd.build_fm_index.preprocess_file (reference scripts/build_fm_index.py:28-73): both corpus formats, the wiki
    markers, empty texts, titles, lower-casing"""
    from seal_amd.build_fm_index import preprocess_file
    kilt = tmp_path / "c.tsv"
    kilt.write_text("p1\tFirst  Title \tSome   text BULLET::::here\n"
                    "p2\tEmpty\t   \n"
                    "bad line without tabs\n"
                    "p3\tThird\tSECTION::::Mixed\tCase\tTabs stay in the text\n")
    labels = []
    got = list(preprocess_file(str(kilt), labels, "kilt", include_title=True))
    assert labels == ["p1", "p3"]
    assert got == ["First  Title @@ Some text here", "Third @@ Mixed Case Tabs stay in the text"]
    labels = []
    assert list(preprocess_file(str(kilt), labels, "kilt", lowercase=True)) == ["some text here", "mixed case tabs stay in the text"]
    labels = []
    assert list(preprocess_file(str(kilt), labels, "kilt", include_title=True, delim="||", word_tokenize=lambda s: s.replace(".", " .").split())) \
        == ["First Title || Some text here", "Third || Mixed Case Tabs stay in the text"]
    dpr = tmp_path / "psgs.tsv"
    dpr.write_text('id\ttext\ttitle\n1\t"a ""quoted"" passage"\tTitle One\n2\tsecond passage\tTitle Two\n')
    labels = []
    assert list(preprocess_file(str(dpr), labels, "dpr", include_title=True)) == ['Title One @@ a "quoted" passage', "Title Two @@ second passage"]
    assert labels == ["1", "2"]


def test_index_builder_tokenises_in_worker_processes_with_a_closure_tokenizer(tmp_path, monkeypatch):
    """``--jobs N``: make_tokenizer() returns a closure (not picklable); the workers must inherit it through fork.
    The index itself is stubbed (no GPU here): the worker fan-out is what is under test."""
    from seal_amd import build_fm_index as b

    class Collect:
        def initialize(self, sequences):
            self.docs = list(sequences)
    monkeypatch.setattr(b, "FMIndex", Collect)
    corpus = tmp_path / "c.tsv"
    corpus.write_text("".join(f"d{i}\tt{i}\tw{i} w{i + 1} w{i + 2}\n" for i in range(700)))
    offset = 10

    def tok(text):                                   # a local closure, like the HF / fairseq tokenizers of make_tokenizer
        return [offset + int(w[1:]) for w in text.split() if w[0] in "tw"] + [2]
    one = b.build_index(str(corpus), tok, include_title=True, jobs=1)
    two = b.build_index(str(corpus), tok, include_title=True, jobs=2)
    assert two.docs == one.docs and len(two.docs) == 700 and two.labels == one.labels == [f"d{i}" for i in range(700)]
    assert b._WORKER_TOKENIZE is None


@pytest.mark.gpu
def test_index_builder_cli_round_trip(tmp_path):
    """corpus file -> build_index (toy tokenizer) -> save -> FMIndex.load: documents, labels and counts survive"""
    from seal_amd import FMIndex
    from seal_amd.build_fm_index import build_index
    corpus = tmp_path / "c.tsv"
    words = ["alpha", "beta", "gamma", "delta", "@@", "omega"]
    corpus.write_text("d0\talpha beta\tgamma delta alpha\n" "d1\tomega\tbeta beta gamma\n" "d2\tdelta\talpha\n")
    tok = lambda text: [10 + words.index(w) for w in text.split()] + [2]      # noqa: E731
    ix = build_index(str(corpus), tok, include_title=True)
    assert ix.labels == ["d0", "d1", "d2"] and ix.n_docs == 3
    assert ix.get_doc(1) == tok("omega @@ beta beta gamma")
    assert ix.get_count(tok("beta gamma")[:-1]) == 1 and ix.get_count([11]) == 3
    ix.save(str(tmp_path / "ix"))
    again = FMIndex.load(str(tmp_path / "ix"))
    assert again.labels == ix.labels and again.get_doc(0) == ix.get_doc(0) and again.get_count([11]) == 3
