"""the reference's command line (cli.py:27-359) offline.  `--pretrained` + `--tokenize data/pan_tadeusz.json`
must write the author's data/pan_tadeusz.tokens.json byte for byte (json.dump(..., ensure_ascii=False, indent=2), cli.py:266-271);
`--train` + `--save` must write the resource files the reference's training writes."""
import json
import os
import shutil

import pytest


def test_cli_parser_matches_the_reference_flags(swt):
    from subword_tokenizers_amd import cli

    a = cli.build_parser().parse_args(["-m", "FastBPE", "FastWordPiece", "--pretrained", "pretrained", "--tokenize", "x.json", "-v", "500",
                                      "-b", "y.json", "-c", "--save", "s", "--train", "t.json", "--reset", "r"])
    assert a.model == ["FastBPE", "FastWordPiece"] and a.max_vocab == 500 and a.compare and a.normalize_with == "bert-base-uncased"
    assert cli.build_parser().parse_args(["-m", "NaiveBPE"]).max_vocab == 1000  # cli.py:126
    assert list(cli.TOKENIZERS) == ["NaiveBPE", "NaiveWordPiece", "FastBPE", "FastWordPiece"]  # cli.py:18-23
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(["--tokenize", "x"])  # --model is required


@pytest.fixture
def workdir(tmp_path, ref_dir, monkeypatch):
    """a working directory laid out like the reference's repository root: resources/pretrained/<Model>/, data/*.json"""
    shutil.copytree(os.path.join(ref_dir, "resources"), tmp_path / "resources")
    # the reference ships byte-identical resource files for the Naive and the Fast class of a family (same sha256): the fixture
    # keeps one copy of each
    for fast, naive in (("FastBPE", "NaiveBPE"), ("FastWordPiece", "NaiveWordPiece")):
        if not (tmp_path / "resources" / "pretrained" / naive).exists():
            shutil.copytree(tmp_path / "resources" / "pretrained" / fast, tmp_path / "resources" / "pretrained" / naive)
    os.makedirs(tmp_path / "data")
    for f in ("pan_tadeusz.json", "train-5K.json"):
        shutil.copy(os.path.join(ref_dir, "data", f), tmp_path / "data" / f)
    monkeypatch.chdir(tmp_path)
    return tmp_path


@pytest.mark.gpu
def test_cli_tokenize_reproduces_the_authors_file(swt, native, workdir, corpora, capsys):
    from subword_tokenizers_amd import cli

    assert cli.main(["--model", "FastBPE", "FastWordPiece", "--pretrained", "pretrained", "--tokenize", "data/pan_tadeusz.json"]) == 0
    out = capsys.readouterr().out
    assert "Loaded saved merges and vocab for FastBPE from resources/pretrained/FastBPE" in out
    assert "Loaded tokenizer model(s): FastBPE, FastWordPiece" in out and "Tokenizing input..." in out
    assert "Tokenized output written to data/pan_tadeusz.tokens.json" in out
    gold = corpora["pan_tokens"]
    assert "[FastBPE] %s" % gold["FastBPE"][0] in out and "[FastWordPiece] %s" % gold["FastWordPiece"][988] in out
    got = (workdir / "data" / "pan_tadeusz.tokens.json").read_bytes()
    want = json.dumps({"FastBPE": gold["FastBPE"], "FastWordPiece": gold["FastWordPiece"]}, ensure_ascii=False, indent=2).encode("utf-8")
    assert got == want  # the author's file restricted to these two models, byte for byte
    # all four models (the Naive classes run the reference's didactic Python loops) on a short file: the author's file holds
    # identical lists for the Naive and the Fast class of a family
    with open(workdir / "data" / "short.json", "w", encoding="utf-8") as f:
        json.dump(corpora["pan"][:25], f, ensure_ascii=False)
    assert cli.main(["-m", "NaiveBPE", "NaiveWordPiece", "FastBPE", "FastWordPiece", "--pretrained", "pretrained", "--tokenize", "data/short.json"]) == 0
    got4 = json.loads((workdir / "data" / "short.tokens.json").read_text(encoding="utf-8"))
    assert list(got4) == ["NaiveBPE", "NaiveWordPiece", "FastBPE", "FastWordPiece"]
    assert got4["NaiveBPE"] == got4["FastBPE"] == gold["FastBPE"][:25] and got4["NaiveWordPiece"] == got4["FastWordPiece"] == gold["FastWordPiece"][:25]
    # a plain string instead of a file: printed, nothing written (cli.py:247-248, 266)
    capsys.readouterr()
    assert cli.main(["-m", "FastBPE", "--pretrained", "pretrained", "--tokenize", "Litwo! Ojczyzno moja!"]) == 0
    assert "[FastBPE] %s\n" % gold["FastBPE"][0][:7] in capsys.readouterr().out
    assert not (workdir / "Litwo! Ojczyzno moja!.tokens.json").exists()


@pytest.mark.gpu
def test_cli_train_save_reset_and_benchmark(swt, native, workdir, golden, capsys):
    from subword_tokenizers_amd import cli

    assert cli.main(["--model", "FastBPE", "--train", "data/train-5K.json", "--max_vocab", "1000", "--save", "mine"]) == 0
    out = capsys.readouterr().out
    assert "Training FastBPE with max_vocab=1000 on 5000 examples..." in out and "Saved merges and vocab for FastBPE to resources/mine/FastBPE" in out
    merges = json.loads((workdir / "resources" / "mine" / "FastBPE" / "merges.json").read_text(encoding="utf-8"))
    assert merges == golden("bpe_train5k_1000.json")["merges"]  # the reference's 922 merges
    # the saved directory loads back and tokenizes; benchmark report keeps the reference's labels
    assert cli.main(["-m", "FastBPE", "--pretrained", "mine", "--benchmark", "data/pan_tadeusz.json"]) == 0
    out = capsys.readouterr().out
    assert "Benchmarking FastBPE (pretrained)..." in out and "=== Tokenization Metrics for FastBPE ===" in out and "=== Zipf Distribution Fit ===" in out
    assert cli.main(["-m", "FastBPE", "FastWordPiece", "--pretrained", "pretrained", "--benchmark", "data/pan_tadeusz.json", "--compare"]) == 0
    out = capsys.readouterr().out
    assert "=== Token Sequence Equivalence (FastBPE vs FastWP) ===" in out and "Positional match rate:" in out
    assert cli.main(["-m", "FastBPE", "--reset", "mine"]) == 0
    assert "Reset resources for FastBPE" in capsys.readouterr().out and not (workdir / "resources" / "mine" / "FastBPE").exists()
    assert cli.main(["-m", "FastBPE", "--reset", "mine"]) == 0
    assert "No resources to reset for FastBPE" in capsys.readouterr().out
