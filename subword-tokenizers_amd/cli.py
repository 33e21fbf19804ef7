"""Offline command line of the reference: /root/reference/cli.py:27-359 with the same flags, the
same resource layout (resources/<DIR>/<ModelName>/merges.json | vocab.json, cli.py:203,232) and the same printed lines, on
top of this package's classes.

One thing differs, on purpose: the reference fetches its pre-tokenizer by NAME over the network
(`AutoTokenizer.from_pretrained(args.normalize_with)`, cli.py:163).  Here the BertPreTokenizer split is the class table that
is compiled into libswt_hip.so (probed from the wheel the reference runs on, tools/gen_unicode_tables.py), so the CLI works
with no network and no Hugging Face cache; `--normalize_with` is accepted and must name a BERT-style (uncased) tokenizer.

    python cli.py --model FastBPE FastWordPiece --pretrained pretrained --tokenize data/pan_tadeusz.json
    python cli.py --model FastBPE --train data/train-5K.json --max_vocab 1000 --save my_dir
    python cli.py --model FastBPE --pretrained pretrained --benchmark data/test.json
"""
import argparse
import json
import os
import shutil
from argparse import RawTextHelpFormatter
from functools import partial

from . import metrics
from .tokenizers import FastBPE, FastWP, NaiveBPE, NaiveWP

MyFormatter = partial(RawTextHelpFormatter, max_help_position=70, width=100)

# cli.py:18-23
TOKENIZERS = {
    "NaiveBPE": NaiveBPE,
    "NaiveWordPiece": NaiveWP,
    "FastBPE": FastBPE,
    "FastWordPiece": FastWP,
}


def build_parser():
    parser = argparse.ArgumentParser(
        prog="cli.py",
        description=("Subword Tokenizers CLI (MI355X)\n\n"
                     "A command-line tool to train and/or tokenize text using various subword tokenizers.\n"),
        formatter_class=MyFormatter,
    )
    parser.add_argument("-m", "--model", choices=TOKENIZERS, nargs="+", metavar=("MODEL1", "MODEL2"), required=True,
                        help=("select primary tokenizer model (required) and optional other models for comparison: "
                              f"{', '.join(TOKENIZERS.keys())}"))
    parser.add_argument("--normalize_with", type=str, metavar="HF_TOKENIZER", default="bert-base-uncased",
                        help="accepted for compatibility: the BertPreTokenizer split is built in (no download)")
    parser.add_argument("--train", type=str, metavar="TRAIN_DATA", help="path to .json file used for training (required to enable training)")
    parser.add_argument("--save", type=str, metavar="PATH", help="save training merges/vocab in specified path for later use")
    parser.add_argument("--pretrained", type=str, metavar="PATH", help="load pretrained merges and vocabulary from specified path")
    parser.add_argument("--tokenize", type=str, metavar="TEST_DATA", help="string to tokenize or path to .json file for tokenization")
    parser.add_argument("-v", "--max_vocab", type=int, metavar="INTEGER", default=1_000, help="maximum vocabulary size for training (default: 1000)")
    parser.add_argument("-b", "--benchmark", type=str, metavar="INPUT",
                        help=("benchmark the selected tokenizer(s)\n"
                              "-\tif --pretrained is provided, INPUT is treated as test data for tokenization benchmarking (string or .json)\n"
                              "-\tif --pretrained is not provided, INPUT is treated as training data for benchmarking training performance (must be .json)\n"
                              "-\tuse --compare to evaluate token sequence equivalence between multiple pretrained models"))
    parser.add_argument("-c", "--compare", action="store_true", help="with --pretrained, only run token-sequence equivalence between models")
    parser.add_argument("--reset", type=str, metavar="PATH",
                        help="reset merges/vocabulary for selected models by deleting their specified resources directory")
    return parser


def _load_inputs(arg):
    """cli.py:244-248: a .json file of sentences, or the argument itself as one sentence"""
    if os.path.isfile(arg) and arg.lower().endswith(".json"):
        with open(arg, "r", encoding="utf-8") as f:
            return json.load(f), True
    return [arg], False


def _tokenize_all(tok, inputs):
    """tok.tokenize(text) for every input -- through the device batch call where the class has one"""
    if hasattr(tok, "tokenize_batch") and len(inputs) > 1:
        for text in inputs:
            if not isinstance(text, str):
                raise TypeError("Text must be a string.")
        return tok.tokenize_batch(list(inputs))
    return [tok.tokenize(text) for text in inputs]


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if "uncased" not in args.normalize_with or "bert" not in args.normalize_with:
        parser.error("--normalize_with: only BERT-style uncased normalization (lowercase + BertPreTokenizer) is built in")

    # cli.py:166-188
    if args.reset:
        for model_name in args.model:
            resource_path = os.path.join("resources", args.reset, model_name)
            if os.path.isdir(resource_path):
                shutil.rmtree(resource_path)
                print(f"Reset resources for {model_name}")
            else:
                print(f"No resources to reset for {model_name}")
        return 0

    # cli.py:191-194 (no HF tokenizer object: the split is built in)
    tokenizer_instances = {name: TOKENIZERS[name]() for name in args.model}

    # cli.py:199-210
    if args.pretrained:
        for name, tok in tokenizer_instances.items():
            resource_path = os.path.join("resources", args.pretrained, name)
            tok.load_resources(resource_path)
            print(f"Loaded saved merges and vocab for {name} from {resource_path}")
    print(f"Loaded tokenizer model(s): {', '.join(tokenizer_instances.keys())}")

    # cli.py:215-236
    if args.train:
        with open(args.train, "r", encoding="utf-8") as f:
            corpus = json.load(f)
        for name, tok in tokenizer_instances.items():
            print(f"Training {name} with max_vocab={args.max_vocab} on {len(corpus)} examples...")
            tok.train(corpus, args.max_vocab)
            if args.save:
                resource_path = os.path.join("resources", args.save, name)
                tok.save_resources(resource_path)
                print(f"Saved merges and vocab for {name} to {resource_path}")

    # cli.py:240-272
    if args.tokenize:
        print("Tokenizing input...")
        inputs, from_file = _load_inputs(args.tokenize)
        per_model = {name: _tokenize_all(tok, inputs) for name, tok in tokenizer_instances.items()}
        output = {}
        for i in range(len(inputs)):
            for name in tokenizer_instances:
                tokens = per_model[name][i]
                print(f"[{name}] {tokens}")
                output.setdefault(name, []).append(tokens)
        if from_file:
            out_path = args.tokenize.replace(".json", ".tokens.json")
            with open(out_path, "w", encoding="utf-8") as f:
                json.dump(output, f, ensure_ascii=False, indent=2)
            print(f"Tokenized output written to {out_path}")

    # cli.py:275-353
    if args.benchmark:
        b_arg = args.benchmark
        if args.pretrained:
            test_inputs, _ = _load_inputs(b_arg)
            train_inputs = []
        else:
            if not os.path.isfile(b_arg) or not b_arg.lower().endswith(".json"):
                parser.error("--benchmark requires TRAIN_INPUT to be a valid .json file path")
            with open(b_arg, "r", encoding="utf-8") as f:
                train_inputs = json.load(f)
            test_inputs = []
        model_names = list(tokenizer_instances.keys())
        models = list(tokenizer_instances.values())
        if args.compare and not args.pretrained:
            parser.error("--compare may only be used with --pretrained")
        if args.compare and len(models) < 2:
            parser.error("--compare requires at least two tokenizers")
        head = model_names[0] if len(models) == 1 else f"{model_names[0]} vs {' vs '.join(model_names[1:])} "
        tail = "" if not train_inputs else f"with {len(train_inputs)} training examples"
        sep = " " if len(models) == 1 else ""
        print(f"Benchmarking {head}{sep}{'(pretrained)' if args.pretrained else ''}{tail}...")
        metrics.benchmarks(tokenizer=models[0], max_vocab_size=args.max_vocab, test_corpus=test_inputs, train_corpus=train_inputs,
                           pretrained=bool(args.pretrained), pretrained_path=os.path.join("resources", args.pretrained, model_names[0]) if args.pretrained else "",
                           reference_tokenizers=models[1:], reference_names=model_names[1:], resources_root=os.path.join("resources", args.pretrained) if args.pretrained else "",
                           compare_only=args.compare)
        print()

    # cli.py:356-359
    if args.save:
        for name, tok in tokenizer_instances.items():
            tok.save_resources(os.path.join("resources", args.save, name))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
