"""``python -m seal_amd.search`` -- the retrieval CLI (reference seal/search.py): topics file in, run file out, every
``SEALSearcher`` parameter as an option.  Same option names and defaults; ``--hybrid`` is accepted for command-line
compatibility (the reference parses it and never reads it)."""
import argparse
import random
from itertools import islice

from .data import OutputFormat, TopicsFormat, get_output_writer, get_query_iterator
from .retrieval import SEALSearcher


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="seal_amd.search")
    p.add_argument("--hybrid", default="none", choices=["none", "ensemble", "recall", "recall-ensemble"])
    p.add_argument("--topics", type=str, metavar="topic_name", required=True, help="topics file")
    p.add_argument("--hits", type=int, metavar="num", default=100, help="number of hits per topic")
    p.add_argument("--topics_format", type=str, metavar="format", default=TopicsFormat.DEFAULT.value,
                   help=f"one of {[x.value for x in TopicsFormat]}")
    p.add_argument("--output_format", type=str, metavar="format", default=OutputFormat.TREC.value,
                   help=f"one of {[x.value for x in OutputFormat]}")
    p.add_argument("--output", type=str, metavar="path", help="output file")
    p.add_argument("--max_passage", action="store_true", default=False, help="keep only the best passage of a document")
    p.add_argument("--max_passage_hits", type=int, metavar="num", default=100, help="hits per topic when --max_passage")
    p.add_argument("--max_passage_delimiter", type=str, metavar="str", default="#", help="between document id and passage id")
    p.add_argument("--remove_duplicates", action="store_true", default=False)
    p.add_argument("--debug", action="store_true", help="first 500 topics only")
    p.add_argument("--keep_samples", type=int, default=None, help="a seeded random sample of this many topics")
    p.add_argument("--chunked", type=int, default=0, help="search the topics in chunks of this many (0: all at once)")
    SEALSearcher.add_args(p)
    return p


def run(args, searcher=None) -> int:
    """returns the number of topics written"""
    queries = get_query_iterator(args.topics, TopicsFormat(args.topics_format))
    writer = get_output_writer(args.output, OutputFormat(args.output_format), "w", max_hits=args.hits, tag="SEAL", topics=queries.topics,
                               use_max_passage=args.max_passage, max_passage_delimiter=args.max_passage_delimiter,
                               max_passage_hits=args.max_passage_hits)
    if args.debug:
        queries.order = queries.order[:500]
    if args.keep_samples is not None and args.keep_samples < len(queries.order):
        random.seed(42)
        random.shuffle(queries.order)
        queries.order = queries.order[:args.keep_samples]
    queries.topics = {t: queries.topics[t] for t in queries.order}
    writer.topics = queries.topics
    if searcher is None:
        searcher = SEALSearcher.from_args(args)
    n = 0
    with writer:
        it = iter(queries)
        while True:
            chunk = list(islice(it, args.chunked)) if args.chunked > 0 else list(it)
            if not chunk:
                break
            topic_ids, texts = zip(*chunk)
            for topic_id, hits in zip(topic_ids, searcher.batch_search(list(texts), k=args.hits)):
                writer.write(topic_id, hits)
                n += 1
            if args.chunked <= 0:
                break
    return n


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    print(args)
    run(args)


if __name__ == "__main__":
    main()
