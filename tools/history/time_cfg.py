import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from genozip_amd import synth
from genozip_amd.codec import Engine
import pyoracle as po
E = Engine(device=0); oracle = po.Oracle()
def T(name, f):
    t=time.time(); r=f(); print("%-40s %.2f s" % (name, time.time()-t), flush=True); return r
rows, cols = 1000, 10000
h = synth.u32(77, rows * cols)
dp = (18 + (h % np.uint32(13)) + ((h >> np.uint32(8)) % np.uint32(13))).astype(np.uint8)
raw = dp.tobytes()
lt, tr = T("gpu local_generate 10MB", lambda: E.local_generate(2, raw, cols))
T("oracle local_generate", lambda: oracle.local_generate(2, raw, cols))
for c in (6, 8, 16):
    g = T("gpu compress codec %d 10MB" % c, lambda: E.compress_many([(c, tr)]))
    T("oracle compress codec %d" % c, lambda: oracle.codec_compress(c, tr))
    T("gpu uncompress codec %d" % c, lambda: E.uncompress_many([(c, g[0], len(tr))]))
n = 10 ** 7
h = T("synth.u32 1e7", lambda: synth.u32(78, n))
ni = np.where(h % np.uint32(10) < np.uint32(8), h % np.uint32(100), h % np.uint32(4200)).astype(np.int32)
seg = T("oracle seg array", lambda: oracle.b250_seg_array(ni, 3000))
n2w = [int(x) for x in (synth.u32(79, 1200) % np.uint32(4240))]
piz = T("gpu b250_generate 1e7", lambda: E.b250_generate(seg, 3000, n2w))
T("oracle b250_generate", lambda: oracle.b250_generate(seg, 3000, n2w))
for c in (6, 8):
    E.profile(True, reset=True)
    E.compress_many([(c, tr)])
    E.profile(False)
    print("codec %d kernels:" % c, "  ".join("%s %.1f ms" % (k.replace("k_", ""), v[0]) for k, v in sorted(E.profile_results().items(), key=lambda kv: -kv[1][0])[:6]), flush=True)
