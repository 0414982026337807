import sys, ctypes as C
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from dlsa_amd import bridge
bridge.lib_path=lambda: 'scratch/libsw_dbg.so'
import dlsa_amd, oracle_lib as o
q='AGCCAATAGAGGATTTACGCCGTGTCGCAATTATGTA'; t='AATTATTGTTGAAATGAAGGCGCCAGTTATCGCACGTAATAACCCTGCGGACGTCTGAGCC'
g,err=bridge.find_rust_alignment_simd_low_mem(q,t,dlsa_amd.AlignmentScore(1,-1,-2))
print(g, o.align("lowmem",q,t,-2,1,-1))
