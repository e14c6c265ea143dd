"""Mutation fuzz of the host-side loaders (scene JSON, binary STL) — meant to run under the sanitizer build
(scripts/cpu_sanitize.sh sets CUTRACE_HOST_LIB): every mutated input must come back as a scene or as an error, never as
a crash or a sanitizer report.  usage: python scripts/cpu_fuzz_loader.py [n]"""
import glob, os, random, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = random.Random(12345)
seeds = [open(p, "rb").read() for p in sorted(glob.glob("scene/*.json"))]
ok = bad = 0
d = tempfile.mkdtemp()
for k in range(n):
    b = bytearray(rng.choice(seeds))
    for _ in range(rng.randint(1, 6)):
        op = rng.randint(0, 4)
        if not b:
            break
        i = rng.randrange(len(b))
        if op == 0:
            b[i] = rng.randrange(256)
        elif op == 1:
            del b[i:i + rng.randint(1, 40)]
        elif op == 2:
            b[i:i] = bytes(rng.choice(b'{}[],:"-0123456789.eE \n') for _ in range(rng.randint(1, 12)))
        elif op == 3:
            b = b[:i]
        else:
            j = rng.randrange(len(b))
            b[i:i] = b[j:j + rng.randint(1, 60)]
    try:
        txt = bytes(b).decode("utf-8", errors="replace")
    except Exception:
        continue
    s = ca.HostScene.parse(txt)
    if s.ok:
        ok += 1
    else:
        bad += 1
print(f"json: {n} mutated scenes: {ok} loaded, {bad} rejected, no crash")
# STL: truncated / corrupted copies referenced from a scene file
stl = open("scene/bunny.stl", "rb").read()
scene = open("scene/bunny.json").read()
ok = bad = 0
for k in range(max(50, n // 20)):
    b = bytearray(stl)
    op = rng.randint(0, 3)
    if op == 0:
        b = b[:rng.randrange(len(b))]
    elif op == 1:
        b[80:84] = rng.randrange(1 << 32).to_bytes(4, "little")  # triangle count lies
    elif op == 2:
        for _ in range(20):
            b[rng.randrange(len(b))] = rng.randrange(256)
    else:
        b = b[:84]
    os.makedirs(os.path.join(d, "scene"), exist_ok=True)
    open(os.path.join(d, "scene", "bunny.stl"), "wb").write(bytes(b))
    open(os.path.join(d, "scene", "bunny.json"), "w").write(scene)
    cwd = os.getcwd()
    os.chdir(d)
    try:
        s = ca.HostScene.load("scene/bunny.json")
    finally:
        os.chdir(cwd)
    if s.ok:
        ok += 1
    else:
        bad += 1
print(f"stl: {ok} loaded, {bad} rejected, no crash")
shutil.rmtree(d, ignore_errors=True)
