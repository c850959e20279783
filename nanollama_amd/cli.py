"""Command-line host -- the flags and modes of go/main.go:25-132 over the HIP engine.

    python -m nanollama_amd --model m.gguf --prompt "Hello"            (one shot)
    python -m nanollama_amd --model m.gguf --interactive               (REPL, /quit /exit /info)
    python -m nanollama_amd --model m.gguf --serve --port 8080         (HTTP: GET /, POST /chat, GET /health)
"""
from __future__ import annotations

import argparse
import os
import sys

from .engine import GenParams
from .engine_text import TextEngine
from .gguf import GGUFError, load_gguf
from .model import load_llama_model
from .tokenizer import Tokenizer


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="nanollama", description="nanollama inference on MI355X (GGUF in, text out)")
    ap.add_argument("--model", default="", help="Path to GGUF model file (required)")
    ap.add_argument("--gamma", default="", help="Path to gamma NPZ file (optional, personality)")
    ap.add_argument("--prompt", default="", help="Prompt text (if empty, reads from stdin)")
    ap.add_argument("--max-tokens", type=int, default=256, help="Maximum tokens to generate")
    ap.add_argument("--temp", type=float, default=0.8, help="Sampling temperature (0 = greedy)")
    ap.add_argument("--top-p", type=float, default=0.9, help="Top-p (nucleus) sampling threshold")
    ap.add_argument("--top-k", type=int, default=50, help="Top-k sampling (used when top-p >= 1.0)")
    ap.add_argument("--rep-penalty", type=float, default=1.15, help="Repetition penalty (1.0 = disabled)")
    ap.add_argument("--rep-window", type=int, default=64, help="Repetition penalty lookback window")
    ap.add_argument("--interactive", action="store_true", help="Interactive REPL mode")
    ap.add_argument("--serve", action="store_true", help="Start HTTP chat server")
    ap.add_argument("--port", type=int, default=8080, help="HTTP server port (used with --serve)")
    ap.add_argument("--list-tensors", action="store_true", help="List all tensors and exit")
    ap.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    ap.add_argument("--gpus", type=int, default=1,
                    help="shard the model tensor-parallel over this many GPUs of the node (2, 4 or 8: devices --device .. --device + N - 1; "
                         "the 7.9B tier's configuration).  One process, one engine: --serve and --interactive work unchanged")
    ap.add_argument("--seed", type=int, default=None, help="sampling seed (default: entropy, like the reference)")
    return ap


def estimate_params(cfg) -> int:
    """estimateParams go/main.go:411-425"""
    embed = cfg.vocab_size * cfg.embed_dim
    attn = cfg.embed_dim * (cfg.num_heads * cfg.head_dim) + 2 * cfg.embed_dim * (cfg.num_kv_heads * cfg.head_dim) + \
        cfg.embed_dim * cfg.embed_dim
    mlp = 3 * cfg.embed_dim * cfg.interm_size
    return 2 * embed + cfg.num_layers * (attn + mlp + 2 * cfg.embed_dim) + cfg.embed_dim


def list_tensors(g) -> None:
    for name, info in g.tensors.items():
        dims = ", ".join(str(d) for d in info.dims[:info.ndims])
        print("  %-50s  type=%d  dims=[%s]  %.2f MB" % (name, info.type, dims, info.nbytes() / 1024 / 1024))


def load_engine(args) -> TextEngine:
    print(f"[nanollama] loading {args.model}")
    g = load_gguf(args.model, verbose=True)
    if args.gpus > 1:
        if args.gpus not in (2, 4, 8):
            raise RuntimeError("--gpus must be 1, 2, 4 or 8")
        devs = list(range(args.device, args.device + args.gpus))
        if os.environ.get("NL_GROUP_ONE_DEVICE"):      # test pool: every rank on --device (what a one-GPU box can run)
            devs = [args.device] * args.gpus
        print(f"[nanollama] tensor-parallel over devices {devs} (push all-reduce between the ranks of this process)")
        model = load_llama_model(g, devices=devs, verbose=True)
    else:
        model = load_llama_model(g, device=args.device, verbose=True)
    if args.gamma:
        from .gamma import load_gamma
        try:
            ge = load_gamma(args.gamma)
            if ge.embed_dim != model.config.embed_dim:
                print(f"warning: gamma embed_dim {ge.embed_dim} != model dim {model.config.embed_dim}, skipping", file=sys.stderr)
            else:
                model.set_gamma(ge.indices, ge.values)
                print(f"[nanollama] personality loaded: {ge.num_tokens} tokens modified")
        except ValueError as exc:
            print(f"warning: failed to load gamma: {exc}", file=sys.stderr)
    tok = Tokenizer(g.meta, verbose=True)
    eng = TextEngine(model, tok, rep_penalty=args.rep_penalty, rep_window=args.rep_window, seed=args.seed)
    print("[nanollama] ready — %dM params, %d layers, %d dim" % (estimate_params(model.config) // 1_000_000,
                                                                 model.config.num_layers, model.config.embed_dim))
    return eng


def run_repl(eng: TextEngine, params: GenParams) -> None:
    """runREPL go/main.go:428-459"""
    print("nanollama interactive mode. Type /quit to exit.\n")
    while True:
        try:
            text = input("> ").strip()
        except EOFError:
            return
        if not text:
            continue
        if text in ("/quit", "/exit"):
            return
        if text == "/info":
            c = eng.model.config
            print(f"Model: {c.num_layers} layers, {c.embed_dim} dim, {c.num_heads} heads, {c.num_kv_heads} kv_heads, {c.vocab_size} vocab")
            print("Params: ~%dM" % (estimate_params(c) // 1_000_000))
            print(f"Gamma: {eng.model.gamma is not None}")
            print(f"Flags: qk_norm={c.qk_norm} rope_conjugate={c.rope_conjugate}")
            continue
        eng.generate(text, params, stream=sys.stdout)
        print()


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    if not args.model:
        print("Usage: nanollama --model <path.gguf> [--gamma <path.npz>] [--prompt <text>]", file=sys.stderr)
        build_parser().print_help(sys.stderr)
        return 1
    try:
        if args.list_tensors:
            list_tensors(load_gguf(args.model, verbose=True))
            return 0
        eng = load_engine(args)
    except (GGUFError, OSError, RuntimeError, ImportError) as exc:
        print(f"error: {exc}", file=sys.stderr)
        return 1
    params = GenParams(max_tokens=args.max_tokens, temperature=args.temp, top_p=args.top_p, top_k=args.top_k)
    if args.serve:
        from .serve import run_server
        run_server(eng, params, args.port)
    elif args.interactive:
        run_repl(eng, params)
    elif args.prompt:
        print(eng.generate(args.prompt, params, stream=sys.stdout))
    else:
        print("> ", end="", flush=True)
        for line in sys.stdin:
            text = line.strip()
            if text in ("/quit", "/exit"):
                break
            if text:
                print(eng.generate(text, params, stream=sys.stdout))
                print()
            print("> ", end="", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
