#!/usr/bin/env python3
"""Generates tests/golden/micro_suppress.npz: greedy streams of the HF Whisper architecture on the
synthetic micro weights with transformers' own SuppressTokensLogitsProcessor / SuppressTokensAtBeginLogitsProcessor applied
to every step's scores (REF-mode semantics otherwise, as tools/make_golden.py).  Dev container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from whisper_mojo_amd import WhisperConfig, synth  # noqa: E402
import make_golden as mg  # noqa: E402


@torch.no_grad()
def greedy(m, enc_out, prompt, steps, processors):
    from transformers.modeling_outputs import BaseModelOutput
    eo = BaseModelOutput(last_hidden_state=enc_out)
    toks = list(prompt)
    out = m(decoder_input_ids=torch.tensor([toks]), encoder_outputs=eo, use_cache=True, decoder_position_ids=torch.arange(len(toks))[None])
    past, cur = out.past_key_values, len(toks)
    for i in range(steps + 1):
        scores = out.logits[:, -1].clone()
        ids = torch.tensor([toks])
        for p in processors:
            scores = p(ids, scores)
        nxt = int(scores[0].argmax())
        toks.append(nxt)
        if i == steps:
            break
        out = m(decoder_input_ids=torch.tensor([[nxt]]), encoder_outputs=eo, past_key_values=past, use_cache=True,
                decoder_position_ids=torch.tensor([[cur - 1]]))  # whisper.mojo:217
        past = out.past_key_values
        cur += 1
    return np.asarray(toks, np.int32)


def main():
    from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor
    torch.manual_seed(0)
    cfg = WhisperConfig.micro()
    w = synth.split_weights(cfg, synth.synth_weights(cfg, 0))
    mel = synth.synth_mel(cfg, 1000)
    m = mg.hf_model(cfg, w, ref_mode=True)
    with mg.TanhStemGelu(True):
        enc_out = m.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state
    prompt, steps = [1, 2, 3, 4], 20
    plain = greedy(m, enc_out, prompt, steps, [])
    sup = sorted(set(int(t) for t in plain[4:]))            # everything the plain stream emits
    s1 = greedy(m, enc_out, prompt, steps, [SuppressTokensLogitsProcessor(sup, device="cpu")])
    bsup = [int(s1[4]), int(s1[5])]                          # its first two generated ids
    s2 = greedy(m, enc_out, prompt, steps, [SuppressTokensLogitsProcessor(sup, device="cpu"),
                                            SuppressTokensAtBeginLogitsProcessor(bsup, begin_index=4, device="cpu")])
    assert not set(s1[4:]) & set(sup) and s2[4] not in bsup and not np.array_equal(plain, s1) and not np.array_equal(s1, s2)
    path = os.path.join(ROOT, "tests", "golden", "micro_suppress.npz")
    np.savez_compressed(path, prompt=np.asarray(prompt, np.int32), plain=plain, suppress=np.asarray(sup, np.int32), with_suppress=s1,
                        begin_suppress=np.asarray(bsup, np.int32), with_both=s2)
    print(path, "plain", plain[4:12], "suppressed", s1[4:12], "both", s2[4:12])


if __name__ == "__main__":
    main()
