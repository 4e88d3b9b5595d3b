import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from __graft_entry__ import load_pkg
import oracle_lib as ol
sbn = load_pkg(); ctx = sbn.Context(0)
G = bytes([1]) + bytes(31) + bytes([2]) + bytes(31)
sc = lambda k: k.to_bytes(32, 'little')
kG = [None] + [ol.g1_mul(G, sc(k)) for k in range(1, 9)]
b = ctx.bases_synthetic(4, 0, sc(1), sc(1))
pts = ctx.bases_download(b, 0, 4)
print("synthetic G..4G", [pts[64*i:64*i+64] == kG[i+1] for i in range(4)])
def t(name, scal, want):
    got = ctx.msm_bases(b, b"".join(sc(k) for k in scal))[0]
    print(name, got == want, got.hex()[:16], want.hex()[:16])
t("1000", (1,0,0,0), kG[1]); t("0100", (0,1,0,0), kG[2]); t("0001", (0,0,0,1), kG[4])
t("1100 same bucket", (1,1,0,0), kG[3]); t("0110", (0,1,1,0), kG[5])
t("2000 weight2", (2,0,0,0), kG[2]); t("3000", (3,0,0,0), kG[3]); t("1200 two buckets", (1,2,0,0), kG[5])
os.environ["SBN_MSM_C"] = "7"
t("c=7 1200", (1,2,0,0), kG[5])
