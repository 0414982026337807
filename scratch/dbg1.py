import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as o, dlsa_amd
def chk(q,t,sc=(-2,1,-1),f="lowmem"):
    e=o.align(f,q,t,*sc)
    g,err=dlsa_amd.find_rust_alignment_simd_low_mem(q,t,dlsa_amd.AlignmentScore(sc[1],sc[2],sc[0]))
    ok=(g.query,g.target,g.score,g.max_x,g.max_y)==(e["q"],e["t"],e["score"],e["max_x"],e["max_y"])
    print("OK " if ok else "BAD", len(q),len(t),(g.query,g.score,g.max_x,g.max_y),(e["q"],e["score"],e["max_x"],e["max_y"]))
chk('AGCCAATAGAGGATTTACGCCGTGTCGCAATTATGTA','AATTATTGTTGAAATGAAGGCGCCAGTTATCGCACGTAATAACCCTGCGGACGTCTGAGCC')
for pos in (0,1,2,3,4,5,10,27,28,29,30,31,32,33,40,60,61,62,63,64,65,100,250,255,256,257,300):
    q="G"*pos+"AATTAT"+"GGG"
    chk(q,"AATTAT"+"C"*(2*len(q)+5))
    chk(q,"C"*7+"AATTAT"+"C"*30, f="lowmem")
