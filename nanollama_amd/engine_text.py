"""Text-level generation: Engine.Generate / GenerateQuiet (go/main.go:152-291) = tokenizer + the id-level loop
of nanollama_amd.engine.Engine."""
from __future__ import annotations

import codecs
import sys
from typing import Optional

from .engine import Engine, GenParams
from .model import LlamaModel
from .tokenizer import Tokenizer


class TextEngine:
    def __init__(self, model: LlamaModel, tokenizer: Tokenizer, rep_penalty: float = 1.15, rep_window: int = 64,
                 seed: Optional[int] = None):
        self.model, self.tokenizer = model, tokenizer
        self.ids = Engine(model, eos_id=tokenizer.eos_id, rep_penalty=rep_penalty, rep_window=rep_window, seed=seed)

    def generate(self, prompt: str, p: GenParams, stream=None) -> str:
        """Generate (go/main.go:152): streams pieces to `stream` (stdout in the CLI), prints "[N tokens, X tok/s]"."""
        tokens = self.tokenizer.encode(prompt, True)
        dec = codecs.getincrementaldecoder("utf-8")(errors="replace")
        out = bytearray()

        def on_token(t: int) -> bool:
            # the reference loop runs while len(output) < 8192 (go/main.go:173): the piece that crosses the cap is
            # the last one emitted, and returning True stops the id-level loop right there (no further sample)
            piece = self.tokenizer.decode_token_bytes(t)
            out.extend(piece)
            if stream is not None:
                stream.write(dec.decode(piece))
                stream.flush()
            return len(out) >= 8192

        self.ids.generate_ids(tokens, p, on_token=on_token)
        if stream is not None:
            stream.write(dec.decode(b"", final=True) + "\n")
            if self.ids.last_tokens and self.ids.last_tok_per_s > 0:
                stream.write("[%d tokens, %.1f tok/s]\n" % (self.ids.last_tokens, self.ids.last_tok_per_s))
            stream.flush()
        return bytes(out).decode("utf-8", errors="replace")

    def generate_quiet(self, prompt: str, p: GenParams) -> str:
        """GenerateQuiet (go/main.go:233-291)."""
        return self.generate(prompt, p, stream=None)
