#!/usr/bin/env python3
"""Generates tests/golden/*.npz — the fixtures that PIN the CPU oracle (oracle/whisper_oracle.c).

Runs ONLY in the dev container (needs `transformers`; never on the GPU box, never at test time).  The
reference's own golden (expected_tokens.txt) was produced by HF `WhisperForConditionalGeneration`
(/root/reference/export_weights.py:125-131); the real checkpoint is unobtainable offline, so we build the
same HF architecture from a local config object, load OUR synthetic weights (whisper.mojo_amd/synth.py,
seed-reproducible) through `load_state_dict` using the name mapping of export_weights.py:19-90, and record
its outputs in two modes:

  HF  : unpatched HF semantics (erf GELU, decoder position = sequence index)  -> what made expected_tokens.txt
  REF : the reference's semantics (tanh GELU whisper_tensor.mojo:288-308; incremental decode positions
        start at current_len-1, whisper.mojo:217) emulated by patching the activation and passing
        decoder_position_ids (modeling_whisper.py:748-759).

No logit processors are applied in either mode (whisper.mojo:198,219 use a raw argmax).
Usage: python tools/make_golden.py            # writes tests/golden/{micro,tiny}_{hf,ref}.npz
       python tools/make_golden.py long       # writes the long / wide pins (round 3): tests/golden/long_*.npz —
                                              #   tiny, REF mode, mel seeds 1000 / 1001 / 1017, the reference's full 195 loop iterations
                                              #   (whisper.mojo:205); tiny, seed 1005, 444 iterations = the 448-row context edge
                                              #   (whisper.mojo:193); tiny HF mode 195 iterations; Whisper-base dims (d = 512, 8 heads,
                                              #   6 + 6 layers), 40 positions.  Top-8 logits + row sums per position: a few KB each.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_mojo_amd import WhisperConfig  # noqa: E402
from whisper_mojo_amd import synth  # noqa: E402

PROMPT = [50258, 50259, 50359, 50363]


def hf_model(cfg: WhisperConfig, weights: dict, ref_mode: bool):
    from transformers import WhisperConfig as HFConfig
    from transformers import WhisperForConditionalGeneration
    hc = HFConfig(vocab_size=cfg.vocab_size, num_mel_bins=cfg.n_mels, d_model=cfg.d_model,
                  encoder_layers=cfg.n_layers, decoder_layers=cfg.n_layers,
                  encoder_attention_heads=cfg.n_heads, decoder_attention_heads=cfg.n_heads,
                  encoder_ffn_dim=cfg.ffn, decoder_ffn_dim=cfg.ffn, max_source_positions=cfg.n_audio_ctx,
                  max_target_positions=cfg.n_text_ctx,
                  activation_function="gelu_pytorch_tanh" if ref_mode else "gelu",
                  pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1,
                  attn_implementation="eager")
    m = WhisperForConditionalGeneration(hc).eval()
    sd = {}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def attn(dst, src):
        sd[dst + "q_proj.weight"] = T(weights[src + "q.w"]); sd[dst + "q_proj.bias"] = T(weights[src + "q.b"])
        sd[dst + "k_proj.weight"] = T(weights[src + "k.w"])
        sd[dst + "v_proj.weight"] = T(weights[src + "v.w"]); sd[dst + "v_proj.bias"] = T(weights[src + "v.b"])
        sd[dst + "out_proj.weight"] = T(weights[src + "o.w"]); sd[dst + "out_proj.bias"] = T(weights[src + "o.b"])

    def ln(dst, src):
        sd[dst + ".weight"] = T(weights[src + ".w"]); sd[dst + ".bias"] = T(weights[src + ".b"])

    def mlp(dst, src):
        for fc in ("fc1", "fc2"):
            sd[dst + fc + ".weight"] = T(weights[src + fc + ".w"]); sd[dst + fc + ".bias"] = T(weights[src + fc + ".b"])

    e = "model.encoder."
    sd[e + "conv1.weight"] = T(weights["enc.conv1.w"]); sd[e + "conv1.bias"] = T(weights["enc.conv1.b"])
    sd[e + "conv2.weight"] = T(weights["enc.conv2.w"]); sd[e + "conv2.bias"] = T(weights["enc.conv2.b"])
    sd[e + "embed_positions.weight"] = T(weights["enc.pos"])
    for l in range(cfg.n_layers):
        p = f"{e}layers.{l}."
        attn(p + "self_attn.", f"enc.{l}.attn."); ln(p + "self_attn_layer_norm", f"enc.{l}.ln1")
        mlp(p, f"enc.{l}."); ln(p + "final_layer_norm", f"enc.{l}.ln2")
    ln(e + "layer_norm", "enc.ln")
    d = "model.decoder."
    sd[d + "embed_tokens.weight"] = T(weights["dec.tok_emb"]); sd[d + "embed_positions.weight"] = T(weights["dec.pos"])
    for l in range(cfg.n_layers):
        p = f"{d}layers.{l}."
        attn(p + "self_attn.", f"dec.{l}.attn."); ln(p + "self_attn_layer_norm", f"dec.{l}.ln1")
        attn(p + "encoder_attn.", f"dec.{l}.cross."); ln(p + "encoder_attn_layer_norm", f"dec.{l}.lnx")
        mlp(p, f"dec.{l}."); ln(p + "final_layer_norm", f"dec.{l}.ln2")
    ln(d + "layer_norm", "dec.ln")
    sd["proj_out.weight"] = sd[d + "embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in k for k in missing), missing
    return m


class TanhStemGelu:
    """The HF encoder hard-codes erf GELU on the conv stem (modeling_whisper.py:618-619); REF mode needs tanh."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        if self.on:
            import transformers.models.whisper.modeling_whisper as mw
            self._orig = mw.nn.functional.gelu
            mw.nn.functional.gelu = lambda x, approximate="none": self._orig(x, approximate="tanh")

    def __exit__(self, *a):
        if self.on:
            import transformers.models.whisper.modeling_whisper as mw
            mw.nn.functional.gelu = self._orig


@torch.no_grad()
def run_decoder(m, enc_out, forced, n_prompt, ref_mode, greedy_steps):
    """Prefill with forced[:n_prompt] then single-token steps.  If greedy_steps>0 feeds back argmax, else
    teacher-forces forced[n_prompt:].  Returns (tokens, logits[steps, V])."""
    from transformers.modeling_outputs import BaseModelOutput
    eo = BaseModelOutput(last_hidden_state=enc_out)
    toks = list(forced[:n_prompt])
    ids = torch.tensor([toks])
    out = m(decoder_input_ids=ids, encoder_outputs=eo, use_cache=True,
            decoder_position_ids=torch.arange(n_prompt)[None])
    past = out.past_key_values
    rows = [out.logits[0, -1].clone()]
    cur_len = n_prompt
    n_steps = greedy_steps if greedy_steps > 0 else len(forced) - n_prompt
    for i in range(n_steps):
        nxt = int(rows[-1].argmax()) if greedy_steps > 0 else int(forced[n_prompt + i])
        toks.append(nxt)
        pos = cur_len - 1 if ref_mode else cur_len  # whisper.mojo:217 vs modeling_whisper.py:749
        out = m(decoder_input_ids=torch.tensor([[nxt]]), encoder_outputs=eo, past_key_values=past, use_cache=True,
                decoder_position_ids=torch.tensor([[pos]]))
        past = out.past_key_values
        rows.append(out.logits[0, -1].clone())
        cur_len += 1
    if greedy_steps > 0:
        toks.append(int(rows[-1].argmax()))
    return np.asarray(toks, np.int32), torch.stack(rows).numpy()


def topk(logits, k=8):
    idx = np.argsort(-logits, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int32), np.take_along_axis(logits, idx, 1)


@torch.no_grad()
def make(cfg_name: str, cfg: WhisperConfig, ref_mode: bool, steps: int, full: bool):
    flat = synth.synth_weights(cfg, 0)
    w = synth.split_weights(cfg, flat)
    mel = synth.synth_mel(cfg, 1000)
    m = hf_model(cfg, w, ref_mode)
    with TanhStemGelu(ref_mode):
        enc = m.model.encoder(torch.from_numpy(mel)[None], output_hidden_states=True)
    hs = [h[0].numpy() for h in enc.hidden_states]  # [0]=stem+pos, [i]=after block i (last one is post-LN'd)
    enc_out = enc.last_hidden_state
    rng = np.random.default_rng(7)
    forced = np.concatenate([PROMPT, rng.integers(0, cfg.vocab_size, steps)]).astype(np.int32) \
        if cfg.vocab_size > 50363 else \
        np.concatenate([rng.integers(0, cfg.vocab_size, 4), rng.integers(0, cfg.vocab_size, steps)]).astype(np.int32)
    prompt = forced[:4]
    g_toks, g_logits = run_decoder(m, enc_out, prompt, 4, ref_mode, greedy_steps=steps)
    f_toks, f_logits = run_decoder(m, enc_out, forced, 4, ref_mode, greedy_steps=0)
    out = dict(mode=np.array("ref" if ref_mode else "hf"), weight_seed=np.int64(0), mel_seed=np.int64(1000),
               prompt=prompt, greedy_tokens=g_toks, forced_tokens=forced)
    eo = enc_out[0].numpy()
    rows = [0, 1, cfg.n_audio_ctx // 2, cfg.n_audio_ctx - 2, cfg.n_audio_ctx - 1]
    out["enc_rows"] = np.asarray(rows, np.int32)
    out["enc_out_rows"] = eo[rows]
    out["enc_out_rowsum"] = eo.astype(np.float64).sum(1)
    out["enc_out_abssum"] = np.abs(eo.astype(np.float64)).sum(1)
    out["stem_rows"] = hs[0][rows]
    out["stem_rowsum"] = hs[0].astype(np.float64).sum(1)
    for i in range(1, cfg.n_layers):  # hidden_states[n_layers] is the LN'd output in HF
        out[f"enc_block{i}_rows"] = hs[i][rows]
    gi, gv = topk(g_logits)
    fi, fv = topk(f_logits)
    out.update(greedy_top_idx=gi, greedy_top_val=gv, forced_top_idx=fi, forced_top_val=fv,
               greedy_logit_sum=g_logits.astype(np.float64).sum(1), forced_logit_sum=f_logits.astype(np.float64).sum(1),
               forced_logit_slice=f_logits[:, :64].copy())
    if full:
        out.update(enc_out=eo, stem=hs[0], greedy_logits=g_logits, forced_logits=f_logits)
    path = os.path.join(ROOT, "tests", "golden", f"{cfg_name}_{'ref' if ref_mode else 'hf'}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; greedy:", g_toks[:12], "margin min",
          float((gv[:, 0] - gv[:, 1]).min()))


@torch.no_grad()
def make_long(name: str, cfg: WhisperConfig, ref_mode: bool, mel_seed: int, steps: int):
    """Free-running greedy decode of `steps` loop iterations (1 prefill + steps single-token forwards): the ids, and per position
    the top-8 logits, the row sum and the first 16 logits.  Pins the oracle at cache lengths the GPU parity tests lean on."""
    flat = synth.synth_weights(cfg, 0)
    w = synth.split_weights(cfg, flat)
    mel = synth.synth_mel(cfg, mel_seed)
    m = hf_model(cfg, w, ref_mode)
    with TanhStemGelu(ref_mode):
        enc_out = m.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state
    toks, logits = run_decoder(m, enc_out, np.asarray(PROMPT, np.int32), 4, ref_mode, greedy_steps=steps)
    ti, tv = topk(logits)
    eo = enc_out[0].numpy()
    out = dict(mode=np.array("ref" if ref_mode else "hf"), weight_seed=np.int64(0), mel_seed=np.int64(mel_seed),
               dims=np.asarray([cfg.d_model, cfg.n_heads, cfg.n_layers, cfg.ffn, cfg.n_mels, cfg.n_audio_ctx, cfg.n_text_ctx, cfg.vocab_size], np.int32),
               prompt=np.asarray(PROMPT, np.int32), greedy_tokens=toks, top_idx=ti, top_val=tv,
               logit_sum=logits.astype(np.float64).sum(1), logit_head=logits[:, :16].copy(),
               enc_out_rowsum=eo.astype(np.float64).sum(1))
    path = os.path.join(ROOT, "tests", "golden", f"long_{name}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(toks), "ids; min top1-top2 margin", float((tv[:, 0] - tv[:, 1]).min()), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        tiny = WhisperConfig.tiny()
        for seed in (1000, 1001, 1017):
            make_long(f"tiny_ref_s{seed}_195", tiny, True, seed, 195)
        make_long("tiny_hf_s1002_195", tiny, False, 1002, 195)
        make_long("tiny_ref_s1005_edge444", tiny, True, 1005, tiny.n_text_ctx - 4)
        make_long("base_ref_s1000_40", WhisperConfig.base(), True, 1000, 39)
        make_long("base_hf_s1003_40", WhisperConfig.base(), False, 1003, 39)
        sys.exit(0)
    for ref_mode in (False, True):
        make("micro", WhisperConfig.micro(), ref_mode, steps=24, full=True)
        make("tiny", WhisperConfig.tiny(), ref_mode, steps=24, full=False)
