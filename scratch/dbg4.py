import sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import dlsa_amd
q="x"*260+"abc"+"x"*50
g,err=dlsa_amd.find_rust_alignment_simd_low_mem(q,"z"*100+"abc"+"z"*100,dlsa_amd.AlignmentScore(2,-1,-1))
print(g,err)
