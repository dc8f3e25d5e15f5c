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
    """symbol -> id as written in a k2 symbol-table text file: whitespace-separated ``<symbol> <id>`` records, one per
    line, blank lines ignored; both columns must be unique.  A table that never assigns id 0 gets ``<eps>`` there, which is
    what the reference's reader leaves behind too (symbol_table.py:67-69, 75-131)."""
    table: Dict[str, int] = {}
    owner_of_id: Dict[int, str] = {}
    with open(path, encoding="utf-8") as f:
        for lineno, raw in enumerate(f, 1):
            rec = raw.split()
            if not rec:
                continue
            if len(rec) != 2:
                raise ValueError(f"{path}:{lineno}: expected '<symbol> <id>', found {len(rec)} fields")
            symbol, ident = rec[0], int(rec[1])
            if symbol in table or ident in owner_of_id:
                clash = symbol if symbol in table else owner_of_id[ident]
                raise ValueError(f"{path}:{lineno}: '{symbol} {ident}' collides with the entry of '{clash}'")
            table[symbol] = ident
            owner_of_id[ident] = symbol
    if 0 not in owner_of_id:
        table.setdefault("<eps>", 0)
    return table


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
