"""Minimal multi-stream capture shapes with plain torch kernels: which fork / join topology makes hipStreamEndCapture crash?"""
import subprocess, sys
CHILD = r'''
import sys, torch
v = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
main_s = torch.cuda.Stream()
a, b, sa, sb = (torch.cuda.Stream() for _ in range(4))
def work(t, n=3):
    for _ in range(n):
        t = t * 1.0001 + 1.0
    return t
def body():
    cur = torch.cuda.current_stream()
    y0 = work(x)
    outs = []
    if v == "flat2":          # two branches forked from the origin, joined back
        for st in (a, b):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(work(y0))
        for st in (a, b):
            cur.wait_stream(st)
    elif v == "nested1":      # one branch with a nested side branch
        a.wait_stream(cur)
        with torch.cuda.stream(a):
            y = work(y0)
            sa.wait_stream(a)
            with torch.cuda.stream(sa):
                z = work(y)
            y = work(y)
            a.wait_stream(sa)
            outs.append(y + z)
        cur.wait_stream(a)
    elif v == "nested2":      # two branches, each with a nested side branch
        for st, ss in ((a, sa), (b, sb)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                y = work(y0)
                ss.wait_stream(st)
                with torch.cuda.stream(ss):
                    z = work(y)
                y = work(y)
                st.wait_stream(ss)
                outs.append(y + z)
        for st in (a, b):
            cur.wait_stream(st)
    elif v == "nested2_event":  # as nested2, the inner join through an explicit event
        for st, ss in ((a, sa), (b, sb)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                y = work(y0)
                ss.wait_stream(st)
                with torch.cuda.stream(ss):
                    z = work(y)
                    e = torch.cuda.Event(); e.record(ss)
                y = work(y)
                st.wait_event(e)
                outs.append(y + z)
        for st in (a, b):
            cur.wait_stream(st)
    elif v == "flat4":        # four branches from the origin
        for st in (a, b, sa, sb):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(work(y0))
        for st in (a, b, sa, sb):
            cur.wait_stream(st)
    r = outs[0]
    for o in outs[1:]:
        r = r + o
    return r
with torch.cuda.stream(main_s):
    ref = body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print(v, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(v, "replayed, equal:", bool(torch.equal(out, ref)), flush=True)
'''
for v in ["flat2", "flat4", "nested1", "nested2", "nested2_event"]:
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, v], capture_output=True, text=True, timeout=120)
    print("=== %s rc=%d | %s | %s" % (v, p.returncode, p.stdout.strip().replace("\n", " ; "), p.stderr.strip()[-300:].replace("\n", " ; ")), flush=True)
