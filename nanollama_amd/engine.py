"""Generation loop -- mirror of Engine.Generate / GenerateQuiet (go/main.go:143-408).

The prompt positions the reference forwards token by token (go/main.go:160-166)
go through model.prefill in one call; decode applies the in-place repetition penalty
(:177-187), then top-p / top-k / argmax (:191-195, :294-408).  By default the
whole loop -- penalty, sampling, recent window, next Forward -- runs on the device
(nl_sample_decode) with the host owning only the random generator and the EOS
stop; `device_sampling=False` keeps the literal per-token host loop over the
logits the device returns.  The pure-greedy configuration (temp <= 0,
rep_penalty <= 1) takes the chained on-device argmax loop.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .model import LlamaModel


@dataclass
class GenParams:
    """go/main.go:135-140 with the CLI defaults of :29-34."""
    max_tokens: int = 256
    temperature: float = 0.8
    top_p: float = 0.9
    top_k: int = 50


def argmax(logits: np.ndarray, n: int) -> int:
    """go/main.go:400-408 (strict '>': lowest index wins ties; numpy argmax has the same rule)."""
    return int(np.argmax(logits[:n]))


class Engine:
    def __init__(self, model: LlamaModel, eos_id: int = 2, rep_penalty: float = 1.15, rep_window: int = 64,
                 seed: Optional[int] = None, device_sampling: bool = True, sample_chunk: int = 32):
        self.model = model
        self.eos_id = eos_id
        self.rep_penalty = np.float32(rep_penalty)
        self.rep_window = rep_window
        self.rng = np.random.default_rng(seed)
        self.device_sampling = device_sampling   # False: read the logits back and sample on the host every token
        self.sample_chunk = sample_chunk         # tokens per nl_sample_decode call (steps after an EOS are wasted)
        self.greedy_chunk = 64                   # tokens per nl_decode_greedy call (four 16-step graph segments)
        self.last_tok_per_s = 0.0
        self.last_tokens = 0

    # sampleTopK go/main.go:294-343
    def sample_top_k(self, temp: float, top_k: int) -> int:
        logits, vocab = self.model.state.logits, self.model.config.vocab_size
        if temp <= 0:
            return argmax(logits, vocab)
        top_k = min(top_k, vocab)
        order = np.argsort(-logits[:vocab], kind="stable")[:top_k]
        vals = logits[order]
        probs = np.exp(((vals - vals[0]) / np.float32(temp)).astype(np.float64)).astype(np.float32)
        r = np.float32(self.rng.random(dtype=np.float32)) * probs.sum(dtype=np.float32)
        cdf = np.cumsum(probs, dtype=np.float32)
        hit = np.nonzero(r <= cdf)[0]
        return int(order[hit[0]]) if len(hit) else int(order[0])

    # sampleTopP go/main.go:346-398
    def sample_top_p(self, temp: float, top_p: float) -> int:
        logits, vocab = self.model.state.logits, self.model.config.vocab_size
        if temp <= 0:
            return argmax(logits, vocab)
        lg = logits[:vocab]
        p = np.exp(((lg - lg.max()) / np.float32(temp)).astype(np.float64)).astype(np.float32)
        p = p * (np.float32(1.0) / p.sum(dtype=np.float32))
        order = np.argsort(-p, kind="stable")
        cs = np.cumsum(p[order], dtype=np.float32)
        cut = int(np.searchsorted(cs, np.float32(top_p), side="left"))
        if cut >= vocab:
            return int(order[0])
        r = np.float32(self.rng.random(dtype=np.float32)) * cs[cut]
        hit = np.nonzero(r <= cs[: cut + 1])[0]
        return int(order[hit[0]]) if len(hit) else int(order[0])

    def generate_ids(self, tokens: List[int], p: GenParams, on_token=None) -> List[int]:
        """Engine.Generate go/main.go:152-230 on token ids (Encode/Decode stay with the tokenizer).

        on_token(id) is called for every generated non-EOS id; a truthy return value ends the generation after that
        token -- the text layer uses it for the reference's 8192-byte output cap (`len(output) < 8192` in the loop
        condition, go/main.go:173): no further token is sampled, the generator, the recent window and the token
        counter stop where the Go loop's would."""
        m, cfg = self.model, self.model.config
        m.reset()
        # prefill, go/main.go:160-166: the Go loop forwards token after token and stops once pos reaches SeqLen-1;
        # the same positions go through nl_prefill in one call (matrix-core path for Q4_0 / Q8_0 files)
        n_prompt = min(len(tokens), max(cfg.seq_len - 1, 0))
        if n_prompt > 0:
            m.prefill(tokens[:n_prompt])
        pos = n_prompt
        out: List[int] = []
        recent: List[int] = []
        start = time.perf_counter()              # timer starts after prefill, go/main.go:171
        greedy_fast = p.temperature <= 0 and self.rep_penalty <= 1.0
        # (the device sampler's select kernel covers vocabularies up to 131072 entries; larger ones take the host loop)
        device_sampling = (not greedy_fast and self.device_sampling and n_prompt > 0 and self.rep_window <= 1024
                           and cfg.vocab_size <= 131072
                           and p.top_p > 0 and (p.top_p < 1.0 or 1 <= p.top_k <= 1024 or p.temperature <= 0))
        if greedy_fast and p.max_tokens > 0:
            # sample_0 comes from the prefill logits; sample_k (k >= 1) exists iff k < max_tokens, the
            # k-th Forward left pos + k < SeqLen (go/main.go:216) and sample_{k-1} was not EOS (:203).
            nxt = argmax(m.state.logits, cfg.vocab_size)
            out.append(nxt)
            n = min(p.max_tokens - 1, cfg.seq_len - 1 - pos)
            # decoded in chunks (whole 16-step graph segments) so that an EOS or the output cap (on_token -> True:
            # `len(output) < 8192` fails before the next sample, go/main.go:173) ends the device work within one chunk:
            # the timer below and the device's KV / position state stop where the Go loop stops, give or take a chunk
            stop = nxt == self.eos_id or bool(on_token and on_token(nxt))
            while not stop and n > 0:
                ids = m.decode_greedy(nxt, pos, min(n, self.greedy_chunk))
                for t in ids:
                    out.append(t)
                    if t == self.eos_id or (on_token and on_token(t)):
                        stop = True
                        break
                if not ids:
                    break
                nxt, pos, n = ids[-1], pos + len(ids), n - len(ids)
            recent = out[-self.rep_window:] if self.rep_window > 0 else []
        elif device_sampling:
            # the same loop with penalty, sampling and the recent window on the device (nl_sample_decode): no
            # read-back of the logits; the host keeps the generator and applies the EOS stop per chunk
            remaining = p.max_tokens
            while remaining > 0 and pos < cfg.seq_len:
                k = min(self.sample_chunk, remaining)
                rng_state = self.rng.bit_generator.state
                us = self.rng.random(k, dtype=np.float32) if p.temperature > 0 else np.zeros(k, np.float32)
                ids, new_recent = m.sample_decode(pos, k, p.temperature, p.top_p, max(p.top_k, 1), float(self.rep_penalty),
                                                  self.rep_window, us, recent)
                used, stop = len(ids), False
                for i, t in enumerate(ids):
                    if t == self.eos_id or (on_token and on_token(t)):   # EOS, or the output cap was reached with this piece
                        used, stop = i + 1, True
                        break
                if used < k and p.temperature > 0:      # leave the generator where the per-token loop would be
                    self.rng.bit_generator.state = rng_state
                    self.rng.random(used, dtype=np.float32)
                if used < len(ids):
                    for t in ids[:used]:
                        recent.append(t)
                        if len(recent) > self.rep_window:
                            recent = recent[1:]
                else:
                    recent = new_recent
                out.extend(ids[:used])
                pos += used
                remaining -= used
                if stop or len(ids) < k:
                    break
        else:
            for _ in range(p.max_tokens):
                logits = m.state.logits
                if self.rep_penalty > 1.0 and recent:     # go/main.go:177-187, in place
                    for tok in recent:
                        if 0 <= tok < cfg.vocab_size:
                            if logits[tok] > 0:
                                logits[tok] /= self.rep_penalty
                            else:
                                logits[tok] *= self.rep_penalty
                nxt = self.sample_top_p(p.temperature, p.top_p) if p.top_p < 1.0 else \
                    self.sample_top_k(p.temperature, p.top_k)
                recent.append(nxt)
                if len(recent) > self.rep_window:
                    recent = recent[1:]
                out.append(nxt)
                if nxt == self.eos_id:
                    break
                capped = bool(on_token(nxt)) if on_token else False
                m.forward(nxt, pos)
                pos += 1
                if pos >= cfg.seq_len or capped:      # (capped: `len(output) < 8192` fails at the top of the next iteration)
                    break
        elapsed = time.perf_counter() - start
        # the reference counts len(recentTokens), capped by --rep-window (go/main.go:198-200,223)
        self.last_tokens = len(recent)
        self.last_tok_per_s = (len(recent) / elapsed) if elapsed > 0 and recent else 0.0
        return out
