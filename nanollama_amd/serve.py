"""HTTP chat server -- the endpoints and JSON shapes of go/serve.go.

    GET  /        chat UI (a minimal page of our own; the reference embeds go/ui.html)
    POST /chat    {"messages":[{"role","content"}], "temperature", "max_tokens", "top_k"} -> {"response"}
    GET  /health  {"status","params_millions","layers","dim","heads","kv_heads","vocab_size","gamma_loaded"}

One generation at a time (the model handle is single-caller): a lock around GenerateQuiet, as go/serve.go:56,106-108.
"""
from __future__ import annotations

import json
import threading
from dataclasses import replace
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

from .engine import GenParams

UI_HTML = b"""<!doctype html><html><head><meta charset="utf-8"><title>nanollama</title>
<style>body{font-family:sans-serif;max-width:46em;margin:2em auto}#log{white-space:pre-wrap;border:1px solid #ccc;
padding:1em;min-height:12em}input{width:80%}</style></head><body><h3>nanollama on MI355X</h3><div id="log"></div>
<p><input id="msg" placeholder="Say something"><button onclick="send()">Send</button></p><script>
async function send(){const m=document.getElementById('msg');const log=document.getElementById('log');
const text=m.value;if(!text)return;log.textContent+='> '+text+'\\n';m.value='';
const r=await fetch('/chat',{method:'POST',headers:{'Content-Type':'application/json'},
body:JSON.stringify({messages:[{role:'user',content:text}]})});const j=await r.json();
log.textContent+=j.response+'\\n\\n';}
document.getElementById('msg').addEventListener('keydown',e=>{if(e.key==='Enter')send();});
</script></body></html>"""


def handle_chat(engine, defaults: GenParams, lock: threading.Lock, body: bytes):
    """POST /chat (go/serve.go:69-111).  Returns (status, payload)."""
    try:
        req = json.loads(body.decode("utf-8") or "{}")
        if not isinstance(req, dict):
            raise ValueError("request must be a JSON object")
    except (ValueError, UnicodeDecodeError) as exc:
        return 400, "bad request: " + str(exc)
    messages = req.get("messages") or []
    if len(messages) == 0:
        return 200, {"response": "Send a message."}
    last = messages[-1] if isinstance(messages[-1], dict) else {}
    prompt = last.get("content") or ""
    if prompt == "":
        return 200, {"response": "Empty message."}
    params = defaults                       # request values override defaults only when > 0 (:94-103)
    if isinstance(req.get("max_tokens"), (int, float)) and req["max_tokens"] > 0:
        params = replace(params, max_tokens=int(req["max_tokens"]))
    if isinstance(req.get("temperature"), (int, float)) and req["temperature"] > 0:
        params = replace(params, temperature=float(req["temperature"]))
    if isinstance(req.get("top_k"), (int, float)) and req["top_k"] > 0:
        params = replace(params, top_k=int(req["top_k"]))
    with lock:
        result = engine.generate_quiet(prompt, params)
    return 200, {"response": result}


def handle_health(engine):
    """GET /health (go/serve.go:114-126)."""
    from .cli import estimate_params
    c = engine.model.config
    return {"status": "ok", "params_millions": estimate_params(c) // 1_000_000, "layers": c.num_layers,
            "dim": c.embed_dim, "heads": c.num_heads, "kv_heads": c.num_kv_heads, "vocab_size": c.vocab_size,
            "gamma_loaded": engine.model.gamma is not None}


def make_handler(engine, defaults: GenParams):
    lock = threading.Lock()

    class Handler(BaseHTTPRequestHandler):
        def log_message(self, fmt, *args):  # quiet
            pass

        def _json(self, status, payload):
            data = (json.dumps(payload) + "\n").encode("utf-8")
            self.send_response(status)
            self.send_header("Content-Type", "application/json")
            self.send_header("Content-Length", str(len(data)))
            self.end_headers()
            self.wfile.write(data)

        def _text(self, status, text):
            data = (text + "\n").encode("utf-8")
            self.send_response(status)
            self.send_header("Content-Type", "text/plain; charset=utf-8")
            self.send_header("Content-Length", str(len(data)))
            self.end_headers()
            self.wfile.write(data)

        def do_GET(self):
            if self.path == "/":
                self.send_response(200)
                self.send_header("Content-Type", "text/html; charset=utf-8")
                self.send_header("Content-Length", str(len(UI_HTML)))
                self.end_headers()
                self.wfile.write(UI_HTML)
            elif self.path == "/health":
                self._json(200, handle_health(engine))
            elif self.path == "/chat":
                self._text(405, "POST only")
            else:
                self._text(404, "404 page not found")

        def do_POST(self):
            if self.path != "/chat":
                self._text(404, "404 page not found")
                return
            n = int(self.headers.get("Content-Length") or 0)
            status, payload = handle_chat(engine, defaults, lock, self.rfile.read(n))
            if isinstance(payload, str):
                self._text(status, payload)
            else:
                self._json(status, payload)

    return Handler


def run_server(engine, defaults: GenParams, port: int) -> None:
    """runServer go/serve.go:54-131"""
    srv = ThreadingHTTPServer(("", port), make_handler(engine, defaults))
    print(f"[nanollama] web UI at http://localhost:{port}")
    try:
        srv.serve_forever()
    finally:
        srv.server_close()
