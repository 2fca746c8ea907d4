"""Why "keep a tile's marginals inside one L2" (VERDICT r3 task 5) cannot bring the check pass of the large codes to its compulsory bytes:
how many of a check pass's marginal-line gathers find their line among the lines touched by the last W gathers, W = the lines half an L2
(4 MB per XCD) holds -- for the edge order in use and for a locality-improving order of the checks (reverse Cuthill-McKee on H H^T).  The ceiling
is 1 - n/E (every line must come from HBM once).  Host only:  python tools/marginal_reuse.py > profiles/r04_marginal_reuse.txt"""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_code
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
def analyse(name, line_bytes, l2_bytes=4<<20):
    code=load_code(name); n,m,E=code.n,code.m,code.E
    chk=np.asarray(code.chk if hasattr(code,'chk') else code.edge_chk); var=np.asarray(code.var if hasattr(code,'var') else code.edge_var)
    cap=l2_bytes//line_bytes   # marginal lines an L2 holds (ignoring the streaming c2v lines that pass through)
    def reuse(order_of_check):
        # process checks in given order; LRU-free model: panel = consecutive checks until distinct variables exceed cap/2 (half the L2 for marginals)
        rank=np.empty(m,int); rank[order_of_check]=np.arange(m)
        o=np.argsort(rank[chk],kind='stable'); v=var[o]
        # sliding model: a gather hits if the same variable was touched within the last W gathers, W chosen s.t. distinct lines ~ cap/2
        last={}; hits=0; W=cap//2
        lastpos=np.full(n,-10**9)
        for i,x in enumerate(v):
            if i-lastpos[x]<=W: hits+=1
            lastpos[x]=i
        return hits/len(v)
    ident=np.arange(m)
    H=sp.csr_matrix((np.ones(E),(chk,var)),shape=(m,n))
    G=(H@H.T).tocsr()
    t=time.time(); rcm=reverse_cuthill_mckee(G,symmetric_mode=True); t=time.time()-t
    print(name,'n',n,'E',E,'lines in half an L2',cap//2,'reuse identity %.3f'%reuse(ident),'RCM %.3f (%.1fs)'%(reuse(rcm),t), 'max possible %.3f'%(1-n/E))
analyse('gen:irg:10000',512)   # fp64, 64-frame tile: 512-byte lines
analyse('gen:irg:10000',256)   # fp32
analyse('gen:reg:64800:3:6',256)
