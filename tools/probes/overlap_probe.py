"""probe: do the E-step (memory-request-bound gathers) and the insertion (instruction / LDS-bound) overlap when they run on
two streams?  Times expectation(half 1) alone, insertion(half 0) alone and both together -- on plain streams and on
streams with complementary CU masks (hipExtStreamCreateWithCUMask)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from thunder_amd.refine import RefineShard
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sh = RefineShard(256, n, dev, batch=2048)
sh.run(1)
sh.refresh_rows(0); sh.refresh_rows(1)
wR, wT = sh.expectation(0)
rot, tran = sh.draw_reco(0, wR, wT)
torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits: words[b // 32] |= (1 << (b % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

def run(s1, s2, label):
    def E():
        with torch.cuda.stream(s1): sh.expectation(1)
    def I():
        with torch.cuda.stream(s2): sh.insertion(0, rot, tran)
    def both():
        E(); I()
    E(); I(); torch.cuda.synchronize()
    tE, tI, tB = timed(E), timed(I), timed(both)
    print("%-28s E alone %.1f ms, I alone %.1f ms, sum %.1f, concurrent %.1f ms (%.2f of the sum)" % (label, tE, tI, tE + tI, tB, tB / (tE + tI)), flush=True)

run(torch.cuda.Stream(), torch.cuda.Stream(), "plain streams")
for x in (64, 96, 128, 160):
    for wg in (2, 4):
        sh.wg_per_cu = wg
        run(masked_stream(range(0, x)), masked_stream(range(x, 256)), "E on %d CUs (cap %d), I on %d" % (x, wg, 256 - x))
