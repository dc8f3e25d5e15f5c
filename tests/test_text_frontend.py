"""CPU: the text front-end wire format against the reference's own classes (imported in the build container;
skipped where /root/reference is absent) and against a committed golden id map."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from oracle.ref_harness import load_reference, reference_available
from valle_amd.text import TextTokenCollater, get_text_token_collater, read_symbol_table, write_symbol_table

SYMS = {"<eps>": 0, "a": 1, "ʃ": 2, "b": 3, "ɔː": 4, "_": 5, ",": 6, "ð": 7}
TEXTS = [["a", "ʃ", "_", "b"], ["ð"], ["ɔː", ",", "a", "a", "b", "_"]]


def test_symbol_table_roundtrip(tmp_path):
    p = str(tmp_path / "unique_text_tokens.k2symbols")
    write_symbol_table(p, SYMS)
    assert read_symbol_table(p) == SYMS
    assert open(p, encoding="utf-8").read().splitlines()[:2] == ["<eps> 0", "a 1"]


def test_collater_matches_committed_golden(tmp_path):
    p = str(tmp_path / "t.k2symbols")
    write_symbol_table(p, SYMS)
    tokens, lens = get_text_token_collater(p)(TEXTS)
    g = json.load(open(os.path.join(GOLDEN, "text_collater.json")))
    assert tokens.tolist() == g["tokens"] and lens.tolist() == g["lens"]
    assert tokens.dtype == torch.int64 and lens.dtype == torch.int32  # collation.py:100-111
    assert tokens[0, 0] == 1 and tokens[1, 2] == 2 and tokens[1, 3] == 0  # <bos>=1, <eos>=2, <pad>=0


@pytest.mark.skipif(not reference_available(), reason="needs /root/reference (build container)")
def test_collater_matches_reference_classes(tmp_path):
    load_reference()
    import importlib
    import sys
    import types

    # valle.utils imports icefall.utils (stubbed by the harness); valle.data is a stub package -> load collation.py by path
    st = importlib.import_module("valle.utils.symbol_table")
    utils = sys.modules.get("valle.utils") or importlib.import_module("valle.utils")
    spec = importlib.util.spec_from_file_location("ref_collation", "/root/reference/valle/data/collation.py")
    col = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(col)
    p = str(tmp_path / "t.k2symbols")
    write_symbol_table(p, SYMS)
    assert st.SymbolTable.from_file(p)._sym2id == read_symbol_table(p)
    rt, rl = col.get_text_token_collater(p)(TEXTS)
    tokens, lens = get_text_token_collater(p)(TEXTS)
    assert torch.equal(rt, tokens) and torch.equal(rl, lens)
    golden = os.path.join(GOLDEN, "text_collater.json")
    if os.environ.get("WRITE_GOLDEN"):
        json.dump({"tokens": rt.tolist(), "lens": rl.tolist()}, open(golden, "w"))
