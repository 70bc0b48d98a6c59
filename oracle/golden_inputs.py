"""Deterministic weights / batch / probes of the d_model = 512 golden case (tests/golden/model_mfma_d512.npz).

TEST INFRASTRUCTURE - never imported by the product package.

The model at the reference's default width (transformer_official.py:115-122: d_model 512, 8 heads x 64, ff 1024) has
7.7 M parameters per encoder + decoder layer pair - too much to commit.  So weights and inputs are a pure function of
numpy seeds (numpy.random.RandomState streams are stable across numpy versions by contract): oracle/gen_golden.py builds
them, loads them into the REFERENCE with load_state_dict, and commits only what the reference computed from them (outputs,
loss, sampled gradient elements, checksums); the tests rebuild the same weights and compare.
Init scales follow the reference (attention.py:16-28, transformer_official.py:147-156, 242-256) so activations have
realistic magnitudes; LayerNorm gains / biases and linear biases are perturbed so the tests see them.
"""
import math

import numpy as np

MFMA_CASE = dict(name="model_mfma_d512", B=3, T=140, F=80, V=60, Lmax=9, wave_len=[140, 97, 71], tgt_len=[9, 4, 6], seed=20241, warm_up=25,
                 cfg=dict(n_mels=80, lfr_m=1, d_model=512, hidden_size=64, ff_size=1024, num_head=8, dropout=0.0, layer_num=1))
SAMPLES = 2048       # gradient elements kept per parameter tensor


def mfma_state_dict(case=MFMA_CASE):
    """name -> float32 numpy array, keys / shapes of TransformerOffical.state_dict() minus the two positional-encoding buffers
    (formula-defined: the reference keeps its own) - load with strict=False or add them."""
    c = case["cfg"]
    d, dk, H, ff, V = c["d_model"], c["hidden_size"], c["num_head"], c["ff_size"], case["V"]
    d_in = c["n_mels"] * c["lfr_m"]
    rs = np.random.RandomState(case["seed"])
    sd = {}

    def normal(shape, std):
        return (rs.standard_normal(shape) * std).astype(np.float32)

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return rs.uniform(-b, b, shape).astype(np.float32)

    def ln(pre):
        sd[pre + "layer_norm.weight"] = (1.0 + 0.1 * rs.standard_normal(d)).astype(np.float32)
        sd[pre + "layer_norm.bias"] = (0.05 * rs.standard_normal(d)).astype(np.float32)

    def mha(pre):
        for n in ("w_qs", "w_ks", "w_vs"):
            sd[pre + n + ".weight"] = normal((H * dk, d), math.sqrt(2.0 / (d + dk)))
            sd[pre + n + ".bias"] = uni((H * dk,), d)
        ln(pre)
        sd[pre + "fc.weight"] = normal((d, H * dk), math.sqrt(2.0 / (d + H * dk)))
        sd[pre + "fc.bias"] = uni((d,), H * dk)

    def ffn(pre):
        sd[pre + "w_1.weight"] = uni((ff, d, 1), d)
        sd[pre + "w_1.bias"] = uni((ff,), d)
        sd[pre + "w_2.weight"] = uni((d, ff, 1), ff)
        sd[pre + "w_2.bias"] = uni((d,), ff)
        ln(pre)

    sd["encoder.linear_in.weight"] = normal((d, d_in), math.sqrt(2.0 / (d + d_in)))
    sd["encoder.linear_in.bias"] = uni((d,), d_in)
    sd["encoder.layer_norm_in.weight"] = (1.0 + 0.1 * rs.standard_normal(d)).astype(np.float32)
    sd["encoder.layer_norm_in.bias"] = (0.05 * rs.standard_normal(d)).astype(np.float32)
    for i in range(c["layer_num"]):
        mha(f"encoder.layer_stack.{i}.slf_attn.")
        ffn(f"encoder.layer_stack.{i}.pos_ffn.")
    # nn.Embedding default is N(0, 1); scaled down so that the tied output projection's logits (no output scaling in the
    # reference, transformer_official.py:321) stay O(1) and the softmax is not saturated
    sd["decoder.tgt_word_emb.weight"] = normal((V, d), 0.05)
    for i in range(c["layer_num"]):
        mha(f"decoder.layer_stack.{i}.slf_attn.")
        mha(f"decoder.layer_stack.{i}.enc_attn.")
        ffn(f"decoder.layer_stack.{i}.pos_ffn.")
    sd["decoder.tgt_word_prj.weight"] = sd["decoder.tgt_word_emb.weight"]
    return sd


def mfma_batch(case=MFMA_CASE):
    """The reference's batch contract (ai_shell_1.py:75-88): zero-padded wave (B, T, F) f32, tgt_for_input (B, Lmax) i64, lengths."""
    rs = np.random.RandomState(case["seed"] + 1)
    B, T, F, V, Lmax = case["B"], case["T"], case["F"], case["V"], case["Lmax"]
    wave = rs.standard_normal((B, T, F)).astype(np.float32)
    tgt = np.zeros((B, Lmax), dtype=np.int64)
    for b in range(B):
        wave[b, case["wave_len"][b]:] = 0.0
        tgt[b, :case["tgt_len"][b]] = rs.randint(4, V, case["tgt_len"][b])
    return dict(wave=wave, tgt_for_input=tgt, wave_len=np.array(case["wave_len"], dtype=np.int64), tgt_len=np.array(case["tgt_len"], dtype=np.int64))


def sample_index(name, numel, case=MFMA_CASE):
    """Sorted flat indices of the elements of parameter `name` whose gradients the golden file keeps (all of them for small tensors)."""
    if numel <= SAMPLES:
        return np.arange(numel)
    h = sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 100003
    rs = np.random.RandomState(case["seed"] + 7 + h)
    return np.sort(rs.choice(numel, SAMPLES, replace=False))


def probe(name, numel, case=MFMA_CASE):
    """A fixed N(0,1) float64 vector per tensor: <gradient, probe> is a checksum over EVERY element."""
    h = sum((i + 3) * ord(ch) for i, ch in enumerate(name)) % 100019
    return np.random.RandomState(case["seed"] + 11 + h).standard_normal(numel)
