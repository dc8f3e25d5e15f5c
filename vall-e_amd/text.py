"""Text front-end wire format of the hot path's caller — host code only.

``bin/infer.py:129-131, 215-221`` turns phoneme strings into the ``x`` / ``x_lens`` tensors that
``VALLE.inference`` consumes with ``get_text_token_collater(args.text_tokens)``; this module
reads the same ``unique_text_tokens.k2symbols`` file (k2 symbol-table text format,
/root/reference/valle/utils/symbol_table.py:75-131: one ``<symbol> <id>`` pair per line, id 0 =
``<eps>`` unless the file says otherwise) and produces the same ids
(/root/reference/valle/data/collation.py:46-54: ``<pad>``=0, ``<bos>``=1, ``<eos>``=2, then the
table's symbols sorted as strings — the file's own ids are NOT used for the model input).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch


def read_symbol_table(path: str) -> Dict[str, int]:
    """symbol -> id as written in the file (symbol_table.py:75-131)."""
    sym2id: Dict[str, int] = {}
    seen_ids = set()
    with open(path, "r", encoding="utf-8") as f:
        for line in f.read().strip().split("\n"):
            fields = line.split()
            if not fields:
                continue
            assert len(fields) == 2, f"Expect a line with 2 fields. Given: {len(fields)}"
            sym, idx = fields[0], int(fields[1])
            assert sym not in sym2id, f"Duplicated symbol {sym}"
            assert idx not in seen_ids, f"Duplicated id {idx}"
            sym2id[sym] = idx
            seen_ids.add(idx)
    if 0 not in seen_ids:  # symbol_table.py:67-69
        sym2id["<eps>"] = 0
    return sym2id


def write_symbol_table(path: str, sym2id: Dict[str, int]) -> None:
    """symbol_table.py:144-163: lines sorted by id."""
    with open(path, "w", encoding="utf-8") as f:
        for sym, idx in sorted(sym2id.items(), key=lambda kv: kv[1]):
            print(sym, idx, file=f)


class TextTokenCollater:
    """Same mapping and padding as the reference's class (collation.py:10-113)."""

    def __init__(self, text_tokens: Sequence[str], add_eos: bool = True, add_bos: bool = True, pad_symbol: str = "<pad>",
                 bos_symbol: str = "<bos>", eos_symbol: str = "<eos>"):
        self.pad_symbol, self.bos_symbol, self.eos_symbol = pad_symbol, bos_symbol, eos_symbol
        self.add_eos, self.add_bos = add_eos, add_bos
        unique = [pad_symbol] + ([bos_symbol] if add_bos else []) + ([eos_symbol] if add_eos else []) + sorted(text_tokens)
        self.token2idx = {t: i for i, t in enumerate(unique)}  # a duplicate keeps its LAST index, as in the reference
        self.idx2token = list(unique)

    def __call__(self, texts: List[Sequence[str]]) -> Tuple[torch.Tensor, torch.Tensor]:
        seqs = [list(t) for t in texts]
        max_len = max(len(s) for s in seqs)
        rows = []
        for s in seqs:
            row = ([self.bos_symbol] if self.add_bos else []) + s + ([self.eos_symbol] if self.add_eos else [])
            row += [self.pad_symbol] * (max_len - len(s))
            rows.append([self.token2idx[t] for t in row])
        lens = [len(s) + int(self.add_eos) + int(self.add_bos) for s in seqs]
        return torch.tensor(rows, dtype=torch.int64), torch.tensor(lens, dtype=torch.int32)


def get_text_token_collater(text_tokens_file: str) -> TextTokenCollater:
    """collation.py:116-122."""
    return TextTokenCollater(sorted(read_symbol_table(text_tokens_file)), add_bos=True, add_eos=True)
