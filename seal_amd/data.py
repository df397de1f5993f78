"""Topic readers and run writers of the search CLI (reference seal/data.py, which builds on pyserini's
``QueryIterator`` / ``OutputWriter`` classes; pyserini is not a dependency here, the few behaviours it contributes --
iteration order, ``max_hits``, max-passage grouping, the TREC / MS MARCO line formats -- are restated below).

Formats keep the reference's names and on-disk shapes (seal/data.py:21-36):

topics   ``default``        TSV ``id<TAB>query`` (pyserini's default reader for a ``.tsv`` file) or a JSON dict
                            ``{id: {"title": query}}``
         ``kilt``           JSON lines with ``id`` and ``input`` (KILT)
         ``kilt_template``  the same files, query = ``meta.template_questions[0]`` (seal/data.py:75-78)
         ``dpr``            a JSON list of ``{"question", ...}`` (seal/data.py:38-52)
         ``dpr_qas``        TSV ``question<TAB>["answer", ...]`` (seal/data.py:54-73)
         ``nq``             JSON lines with ``example_id`` / ``question_text`` (seal/data.py:80-95; the reference never fills the
                            iteration order of this reader, so it iterates over nothing -- here the file order is used)
output   ``trec``           ``topic Q0 docid rank score tag``
         ``msmarco``        ``topic<TAB>docid<TAB>rank``
         ``kilt``           one JSON object per topic with a provenance list (seal/data.py:110-140)
         ``dpr``            the DPR topics with a ``ctxs`` list added, one JSON document at the end (seal/data.py:142-166)
"""
import ast
import csv
import json
import re
from enum import Enum, unique
from typing import Dict, Iterator, List, Optional, Tuple


@unique
class TopicsFormat(Enum):
    DEFAULT = "default"
    KILT = "kilt"
    KILT_TEMPLATE = "kilt_template"
    DPR = "dpr"
    DPR_QAS = "dpr_qas"
    NQ = "nq"


@unique
class OutputFormat(Enum):
    TREC = "trec"
    MSMARCO = "msmarco"
    KILT = "kilt"
    DPR = "dpr"


class QueryIterator:
    """``topics`` (id -> record) + ``order`` (ids); iterating yields ``(id, query text)`` in that order"""

    def __init__(self, topics: Dict, order: List, field=None):
        self.topics, self.order, self._field = topics, order, field

    def get_query(self, id_):
        return self._field(self.topics[id_])

    def __iter__(self) -> Iterator[Tuple[object, str]]:
        for id_ in self.order:
            yield id_, self.get_query(id_)

    def __len__(self):
        return len(self.order)


def _json_lines(path):
    with open(path) as f:
        for line in f:
            if line.strip():
                yield json.loads(line)


def get_query_iterator(topics_path: str, topics_format: TopicsFormat, queries_path: Optional[str] = None) -> QueryIterator:
    fmt = TopicsFormat(topics_format)
    topics, order = {}, []
    if fmt is TopicsFormat.DEFAULT:
        if str(topics_path).endswith(".json"):
            with open(topics_path) as f:
                for k, v in json.load(f).items():
                    topics[k] = v if isinstance(v, dict) else {"title": v}
                    order.append(k)
        else:
            with open(topics_path, newline="") as f:
                rows = [row for row in csv.reader(f, delimiter="\t", quoting=csv.QUOTE_NONE) if len(row) >= 2]
            # pyserini reads .tsv topics with TsvIntTopicReader: integer topic ids (kept as strings only if some id is not one)
            as_int = all(re.fullmatch(r"-?\d+", row[0].strip()) for row in rows)
            for row in rows:
                topics[int(row[0]) if as_int else row[0]] = {"title": row[1]}
        # pyserini's QueryIterator walks sorted(topics) when the topic set has no predefined order: --debug (first 500)
        # and --keep_samples (seed-42 shuffle of this order) then pick the same topics as the reference CLI
        order = sorted(topics)
        return QueryIterator(topics, order, lambda t: t["title"])
    if fmt in (TopicsFormat.KILT, TopicsFormat.KILT_TEMPLATE):
        for inst in _json_lines(topics_path):
            topics[inst["id"]] = inst
            order.append(inst["id"])
        if fmt is TopicsFormat.KILT:
            return QueryIterator(topics, order, lambda t: t["input"])
        return QueryIterator(topics, order, lambda t: t["meta"]["template_questions"][0])
    if fmt is TopicsFormat.DPR:
        with open(topics_path) as f:
            for i, inst in enumerate(json.load(f)):
                topics[i] = inst
                order.append(i)
        return QueryIterator(topics, order, lambda t: t["question"])
    if fmt is TopicsFormat.DPR_QAS:
        with open(topics_path, newline="") as f:
            for i, (query, answers) in enumerate(csv.reader(f, delimiter="\t", quotechar='"')):
                answers = ast.literal_eval(answers)
                if not (isinstance(answers, list) and answers and isinstance(answers[0], str)):
                    raise ValueError(f"{topics_path}:{i + 1}: the second column must be a list of answer strings")
                topics[i] = {"question": query, "answers": answers}
                order.append(i)
        return QueryIterator(topics, order, lambda t: t["question"])
    for inst in _json_lines(topics_path):           # NQ
        topics[inst["example_id"]] = inst
        order.append(inst["example_id"])
    return QueryIterator(topics, order, lambda t: t["question_text"])


class OutputWriter:
    """context manager writing one topic at a time; ``hits`` are objects with ``docid`` / ``score`` (``SEALDocument``).
    ``use_max_passage``: the part of ``docid`` before ``max_passage_delimiter`` names the document; only the first
    (best) passage of a document is kept, up to ``max_passage_hits`` documents."""

    def __init__(self, file_path: str, mode: str = "w", max_hits: int = 1000, tag: Optional[str] = None, topics: Optional[dict] = None,
                 use_max_passage: bool = False, max_passage_delimiter: Optional[str] = None, max_passage_hits: int = 100):
        self.file_path, self.mode, self.max_hits, self.tag, self.topics = file_path, mode, max_hits, tag, topics
        self.use_max_passage, self.max_passage_delimiter, self.max_passage_hits = use_max_passage, max_passage_delimiter, max_passage_hits
        self._file = None

    def __enter__(self):
        self._file = open(self.file_path, self.mode)
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self._file.close()
        return False

    def hits_iterator(self, hits):
        seen, rank = set(), 1
        for hit in hits:
            if self.use_max_passage and rank > self.max_passage_hits:
                break
            if not self.use_max_passage and rank > self.max_hits:
                break
            docid = str(hit.docid).strip()
            if self.use_max_passage:
                docid = docid.split(self.max_passage_delimiter)[0]
                if docid in seen:
                    continue
                seen.add(docid)
            yield docid, rank, hit.score, hit
            rank += 1

    def write(self, topic, hits):
        raise NotImplementedError


class TrecWriter(OutputWriter):
    def write(self, topic, hits):
        for docid, rank, score, _ in self.hits_iterator(hits):
            self._file.write(f"{topic} Q0 {docid} {rank} {score:.6f} {self.tag}\n")


class MsMarcoWriter(OutputWriter):
    def write(self, topic, hits):
        for docid, rank, _, _ in self.hits_iterator(hits):
            self._file.write(f"{topic}\t{docid}\t{rank}\n")


class KiltWriter(OutputWriter):
    """seal/data.py:110-140: docid ``<wikipedia id>[-<start paragraph>[-<end paragraph>]]``"""

    def write(self, topic, hits):
        provenance = []
        datapoint = {"id": topic, "input": None, "output": [{"provenance": provenance}]}
        for docid, _, score, hit in self.hits_iterator(hits):
            if not hasattr(hit, "text"):
                provenance.append({"wikipedia_id": docid})
                continue
            if datapoint["input"] is None and getattr(hit, "query", None) is not None:
                datapoint["input"] = hit.query
            parts = docid.split("-")
            start = end = 0
            if len(parts) == 2:
                start = end = int(parts[1])
            elif len(parts) >= 3:
                start, end = int(parts[1]), int(parts[2])
            title, body = hit.text()
            provenance.append({"wikipedia_id": int(parts[0]), "start_paragraph_id": start, "end_paragraph_id": end,
                               "text": f"{title} @@ {body}", "score": score})
            if getattr(hit, "keys", None) is not None:
                provenance[-1]["meta"] = {"keys": hit.keys}
        json.dump(datapoint, self._file)
        self._file.write("\n")


class DprWriter(OutputWriter):
    """seal/data.py:142-166: the topics themselves, each with its ``ctxs``, written as one JSON list on exit"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.order = []

    def write(self, topic, hits):
        datapoint = self.topics[topic]
        self.order.append(topic)
        ctxs = datapoint["ctxs"] = []
        for docid, _, score, hit in self.hits_iterator(hits):
            title, body = hit.text()
            ctxs.append({"title": title.strip(), "text": body.strip(), "score": score, "passage_id": docid})

    def __exit__(self, exc_type, exc_value, exc_traceback):
        json.dump([self.topics[t] for t in self.order], self._file, indent="    ")
        return super().__exit__(exc_type, exc_value, exc_traceback)


def get_output_writer(file_path: str, output_format: OutputFormat, *args, **kwargs) -> OutputWriter:
    return {OutputFormat.TREC: TrecWriter, OutputFormat.MSMARCO: MsMarcoWriter, OutputFormat.KILT: KiltWriter,
            OutputFormat.DPR: DprWriter}[OutputFormat(output_format)](file_path, *args, **kwargs)
