"""Host logic above the C ABI that needs no GPU: CLI flags (go/main.go:26-38) and the HTTP handlers of
go/serve.go driven with a stand-in engine."""
import json
import threading
import urllib.request
from http.server import ThreadingHTTPServer
from types import SimpleNamespace

from nanollama_amd import cli, serve
from nanollama_amd.engine import GenParams


class FakeEngine:
    def __init__(self):
        self.calls = []
        self.model = SimpleNamespace(gamma=None, config=SimpleNamespace(
            num_layers=13, embed_dim=576, num_heads=9, num_kv_heads=9, head_dim=64, vocab_size=32000, interm_size=1536,
            qk_norm=False, rope_conjugate=False))

    def generate_quiet(self, prompt, params):
        self.calls.append((prompt, params))
        return "echo:" + prompt


def test_cli_flags_and_defaults_match_the_go_binary():
    a = cli.build_parser().parse_args(["--model", "m.gguf"])
    assert (a.max_tokens, a.temp, a.top_p, a.top_k, a.rep_penalty, a.rep_window, a.port) == (256, 0.8, 0.9, 50, 1.15, 64, 8080)
    assert not (a.interactive or a.serve or a.list_tensors) and a.prompt == "" and a.gamma == ""
    assert cli.main([]) == 1                                        # no --model: usage, exit 1
    assert cli.main(["--model", "/nonexistent.gguf"]) == 1          # LoadGGUF error -> exit 1
    assert cli.estimate_params(FakeEngine().model.config) // 1_000_000 == 88


def test_chat_handler_semantics():
    eng, lock, d = FakeEngine(), threading.Lock(), GenParams()
    st, out = serve.handle_chat(eng, d, lock, b'{"messages":[{"role":"user","content":"a"},{"role":"user","content":"hi"}]}')
    assert (st, out) == (200, {"response": "echo:hi"}) and eng.calls[-1][1] == d      # LAST message is the prompt
    st, out = serve.handle_chat(eng, d, lock, b'{"messages":[{"content":"x"}],"temperature":0.2,"max_tokens":7,"top_k":3}')
    p = eng.calls[-1][1]
    assert (p.temperature, p.max_tokens, p.top_k, p.top_p) == (0.2, 7, 3, d.top_p)
    st, out = serve.handle_chat(eng, d, lock, b'{"messages":[{"content":"x"}],"temperature":0,"max_tokens":-1}')
    assert eng.calls[-1][1] == d                                     # only values > 0 override
    assert serve.handle_chat(eng, d, lock, b'{"messages":[]}') == (200, {"response": "Send a message."})
    assert serve.handle_chat(eng, d, lock, b'{"messages":[{"content":""}]}') == (200, {"response": "Empty message."})
    st, out = serve.handle_chat(eng, d, lock, b'{not json')
    assert st == 400 and out.startswith("bad request")
    h = serve.handle_health(eng)
    assert h == {"status": "ok", "params_millions": 88, "layers": 13, "dim": 576, "heads": 9, "kv_heads": 9,
                 "vocab_size": 32000, "gamma_loaded": False}


def test_http_server_routes():
    eng = FakeEngine()
    srv = ThreadingHTTPServer(("127.0.0.1", 0), serve.make_handler(eng, GenParams()))
    port = srv.server_address[1]
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    try:
        base = f"http://127.0.0.1:{port}"
        assert b"nanollama" in urllib.request.urlopen(base + "/").read()
        assert json.loads(urllib.request.urlopen(base + "/health").read())["layers"] == 13
        req = urllib.request.Request(base + "/chat", data=b'{"messages":[{"role":"user","content":"yo"}]}',
                                     headers={"Content-Type": "application/json"})
        assert json.loads(urllib.request.urlopen(req).read()) == {"response": "echo:yo"}
        for path, code in (("/nope", 404), ("/chat", 405)):
            try:
                urllib.request.urlopen(base + path)
                assert False
            except urllib.error.HTTPError as e:
                assert e.code == code
    finally:
        srv.shutdown()
        srv.server_close()
