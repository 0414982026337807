import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as o, dlsa_amd
from dlsa_amd import bridge
def chk(q,t,sc=(-1,2,-1)):
    e=o.align("lowmem",q,t,*sc)
    g,err=dlsa_amd.find_rust_alignment_simd_low_mem(q,t,dlsa_amd.AlignmentScore(sc[1],sc[2],sc[0]))
    print(len(q),len(t),"GPU",None if g is None else (g.query[:20],g.score,g.max_x,g.max_y),"ERR",err,"EXP",(e["q"][:20],e["score"],e["max_x"],e["max_y"]))
for pos in (10,250,260,300,520,1000):
    q="x"*pos+"abc"+"x"*50
    chk(q,"z"*100+"abc"+"z"*100)
    chk(q,"z"*2000+"abc"+"z"*1000)
