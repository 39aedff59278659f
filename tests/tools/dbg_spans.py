import sys, numpy as np
sys.path.insert(0, '.')
import vfind_amd
from oracle import oracle
from tests import synth
ads = ("GATCGGAAGAGCACACGTCT", "AGATCGGAAGAGCGTCGTGT")
reads = synth.make_reads(7, 4096, ads, L=150, sub=0.02, indel=0.005, dirty=0.0)
with vfind_amd.Engine(ads, device=0, table_slots=1 << 16, arena_bytes=1 << 22) as e:
    st, en = e.spans(reads)
ost, oen = oracle.spans(reads, ads)
bad = np.nonzero((st != ost) | (en != oen))[0]
print("n bad", len(bad), "of", len(reads))
for i in bad[:12]:
    r = reads[i]
    print(i, "gpu", st[i], en[i], "oracle", ost[i], oen[i], "exactP", oracle.find_exact(r, ads[0]), "exactS", oracle.find_exact(r, ads[1]),
          "sgP", oracle.sg_stats(ads[0], r), "sgS", oracle.sg_stats(ads[1], r))
