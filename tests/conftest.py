import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The host oracle at test sizes (d <= 256) is a long chain of tiny matmuls: with the default intra-op pool on a shared box the
# same GPU test took 4.5 s in one run and 42 s in the next (oversubscribed threads); four threads are as fast and steady.
torch.set_num_threads(min(4, torch.get_num_threads()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_names(kind="inference"):
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    if kind == "continual":
        return [n for n in names if n.startswith("continual")]
    if kind == "vallf":  # VALLF.inference fixtures (valle.py:566-710)
        return [n for n in names if n.startswith("vallf")]
    return [n for n in names if not n.startswith("continual") and not n.startswith("vallf")]


class Golden:
    """A committed fixture: reference outputs + everything needed to rebuild its inputs."""

    def __init__(self, name):
        from valle_amd.config import ModelConfig

        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        c = [int(v) for v in z["cfg"]]
        self.cfg = ModelConfig(model_name="VALL-F" if ("vallf" in z and int(z["vallf"])) else "VALL-E", decoder_dim=c[0], nhead=c[1], num_decoder_layers=c[2], prefix_mode=c[3],
                               prepend_bos=bool(c[4]), num_quantizers=c[5], share_embedding=bool(c[6]),
                               scale_factor=float(z["scale_factor"]) if "scale_factor" in z else 1.0,
                               norm_first=bool(int(z["norm_first"])) if "norm_first" in z else True,
                               add_prenet=bool(int(z["add_prenet"])) if "add_prenet" in z else False)
        self.weight_seed = int(z["weight_seed"])
        self.x = torch.from_numpy(z["x"].astype(np.int64))
        self.x_lens = torch.from_numpy(z["x_lens"])
        self.y = torch.from_numpy(z["y"].astype(np.int64))
        self.codes = torch.from_numpy(z["codes"].astype(np.int64))
        if "top_k" not in z:  # VALLE.continual fixtures: inputs + codes only
            return
        e = int(z["enroll"])
        self.enroll_x_lens = None if e < 0 else torch.tensor([e], dtype=torch.int32)
        self.top_k = int(z["top_k"])
        self.temperature = float(z["temperature"])
        self.n_pass = int(z["n_pass"])
        self.ar_probe_steps = [int(v) for v in z["ar_probe_steps"]]
        self.ar_probe_logits = torch.from_numpy(z["ar_probe_logits"])
        self.nar_probe_logits = torch.from_numpy(z["nar_probe_logits"]) if "nar_probe_logits" in z else None
        self.exp_noise = torch.from_numpy(z["exp_noise"]) if "exp_noise" in z else None
        self.sample_seed = int(z["sample_seed"]) if "sample_seed" in z else None

    def state_dict(self):
        from valle_amd.weights import synthetic_state_dict

        return synthetic_state_dict(self.cfg, self.weight_seed)

    def oracle(self, sd=None):
        from oracle import valle_oracle as vo

        c = self.cfg
        if c.is_vallf:
            return vo.OracleModelF(sd if sd is not None else self.state_dict(), c.decoder_dim, c.nhead, c.num_decoder_layers,
                                   prefix_mode=c.prefix_mode, prepend_bos=c.prepend_bos, num_quantizers=c.num_quantizers,
                                   nar_scale_factor=c.scale_factor, norm_first=c.norm_first, add_prenet=c.add_prenet)
        return vo.OracleModel(sd if sd is not None else self.state_dict(), c.decoder_dim, c.nhead,
                              c.num_decoder_layers, c.prefix_mode, c.prepend_bos, c.num_quantizers, c.scale_factor,
                              c.norm_first, c.add_prenet)


@pytest.fixture(scope="session")
def golden_cache():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


class NarStats:
    """tests/golden/narstats/<name>.npz (oracle/gen_nar_stats.py): per-stage logit statistics of the unmodified reference for
    every generated row (top-1 / top-2 values, largest magnitude) and the full logits of a few rows."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, "narstats", name + ".npz"))
        self.top1 = torch.from_numpy(z["top1"])          # (Q-1, T)
        self.top2 = torch.from_numpy(z["top2"])
        self.absmax = torch.from_numpy(z["absmax"])
        self.rows = torch.from_numpy(z["rows"].astype(np.int64))
        self.row_logits = torch.from_numpy(z["row_logits"])  # (Q-1, len(rows), 1024)

    def decided(self, rel_tol):
        """(Q-1, T) bool: rows whose reference decision margin exceeds twice the tolerance (rel_tol x the row's logit scale):
        there the index selection must be exact (north_star: bit-exact argmax wherever the margin allows)."""
        return (self.top1 - self.top2) > 2 * rel_tol * self.absmax


def nar_stats_available(name):
    return os.path.isfile(os.path.join(GOLDEN, "narstats", name + ".npz"))
