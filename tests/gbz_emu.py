"""Pure-Python reader of the GBWT inside a GBZ file (simple-sds layout) -- TEST INFRASTRUCTURE ONLY.

An independent restatement of what pangenome-index_amd/csrc/pgx_gbz.cpp parses (jltsiren/gbwt's published file format;
the library itself is not under /root/reference): StringArray tags, the record array, CompressedRecord edges and runs,
and an LF walk that spells every path as its node sequence.  tests/test_gbz.py checks the walks on the reference's GBZ
fixtures (every path ends at the endmarker after header.size steps; reverse paths mirror forward ones) and compares the
C++ reader with it."""
import struct, sys
class R:
    def __init__(s, d, o=0): s.d=d; s.o=o
    def u64(s):
        v=struct.unpack_from('<Q', s.d, s.o)[0]; s.o+=8; return v
    def vec_u64(s):
        n=s.u64(); v=list(struct.unpack_from('<%dQ'%n, s.d, s.o)); s.o+=8*n; return v
    def vec_u8(s):
        n=s.u64(); v=s.d[s.o:s.o+n]; s.o+=(n+7)//8*8; return v
    def raw(s):
        bits=s.u64(); w=s.vec_u64(); return bits, w
    def option_skip(s):
        n=s.u64(); s.o+=8*n; return n
    def intvec(s):
        n=s.u64(); w=s.u64(); bits, words=s.raw()
        assert bits==n*w, (bits,n,w)
        big=0
        for i,x in enumerate(words): big|=x<<(64*i)
        return [(big>>(i*w))&((1<<w)-1) for i in range(n)]
    def bitvec(s):
        ones=s.u64(); bits, words=s.raw()
        for _ in range(3): s.option_skip()
        big=0
        for i,x in enumerate(words): big|=x<<(64*i)
        return ones, bits, big
    def sparse(s):
        n=s.u64(); ones, hbits, high=s.bitvec(); low=s.intvec()
        w = None
        vals=[]; zeros=0; k=0
        # width of low = from intvec: recompute
        return n, ones, hbits, high, low
def sparse_values(s):
    o0=s.o
    n=s.u64(); ones=s.u64(); hbits=s.u64(); nw=s.u64(); words=[s.u64() for _ in range(nw)]
    for _ in range(3): s.option_skip()
    ln=s.u64(); lw=s.u64(); lbits=s.u64(); lnw=s.u64(); lwords=[s.u64() for _ in range(lnw)]
    high=0
    for i,x in enumerate(words): high|=x<<(64*i)
    lowb=0
    for i,x in enumerate(lwords): lowb|=x<<(64*i)
    vals=[]; zeros=0; k=0
    for pos in range(hbits):
        if (high>>pos)&1:
            vals.append((zeros<<lw) | ((lowb>>(k*lw))&((1<<lw)-1))); k+=1
        else: zeros+=1
    assert k==ones==ln, (k,ones,ln)
    return n, vals
def string_array(s):
    n, offs = sparse_values(s)
    alpha = s.vec_u8()
    chars = s.intvec()
    text = bytes(alpha[c] for c in chars)
    offs2 = offs+[len(text)]
    return [text[offs2[i]:offs2[i+1]] for i in range(len(offs))]
def bytecode(d, o):
    v=0; sh=0
    while True:
        b=d[o]; o+=1
        v|=(b&0x7F)<<sh; sh+=7
        if not b&0x80: return v,o
def parse_gbwt(path):
    d=open(path,'rb').read()
    s=R(d)
    s.u64(); s.u64(); string_array(s)
    hdr=struct.unpack_from('<IIQQQQQ', d, s.o); s.o+=48
    tag,ver,nseq,size,offset,sigma,flags=hdr
    string_array(s)
    n, starts = sparse_values(s)
    data = s.vec_u8()
    assert n == len(data), (n, len(data))
    nrec=len(starts)
    assert nrec == sigma-offset
    recs=[]
    lims=starts+[len(data)]
    for r in range(nrec):
        o, end = lims[r], lims[r+1]
        if o==end: recs.append(([],[])); continue
        outdeg,o=bytecode(data,o)
        edges=[]; prev=0
        for _ in range(outdeg):
            dn,o=bytecode(data,o); off,o=bytecode(data,o)
            prev+=dn; edges.append((prev,off))
        runs=[]
        rc = 256//outdeg if outdeg and outdeg<255 else 0
        while o<end:
            if rc==0:
                rk,o=bytecode(data,o); ln,o=bytecode(data,o); ln+=1
            else:
                c=data[o]; o+=1
                rk, ln = c%outdeg, c//outdeg+1
                if ln>=rc:
                    x,o=bytecode(data,o); ln+=x
            runs.append((rk,ln))
        recs.append((edges,runs))
    return dict(nseq=nseq,size=size,offset=offset,sigma=sigma,recs=recs,flags=flags)

def walk(g, seq):
    """node sequence of GBWT sequence seq"""
    recs, off = g['recs'], g['offset']
    def comp(node): return 0 if node==0 else node-off
    def lf(node, i):
        edges, runs = recs[comp(node)]
        # position i within record: find run
        pos=0; cnt=[e[1] for e in edges]
        for rk,ln in runs:
            if i < pos+ln:
                return edges[rk][0], cnt[rk] + (i-pos)
            cnt[rk]+=ln; pos+=ln
        raise ValueError
    node,i = 0, seq
    out=[]
    node,i = lf(0, seq)
    while node!=0:
        out.append(node)
        node,i=lf(node,i)
    return out


def components(g):
    """weakly connected components over graph node ids (GBWT node // 2), numbered by smallest node id"""
    off = g['offset']
    parent = {}
    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for r, (edges, _runs) in enumerate(g['recs']):
        if r == 0 or not (edges or _runs):
            continue
        node = r + off
        find(node // 2)
        for t, _ in edges:
            if t:
                a, b = find(node // 2), find(t // 2)
                if a != b:
                    parent[max(a, b)] = min(a, b)
    comp, out = {}, {}
    for v in sorted(parent):
        root = find(v)
        if root not in comp:
            comp[root] = len(comp)
        out[v] = comp[root]
    return out
