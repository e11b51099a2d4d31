import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'pangenome-index_amd')
import numpy as np
import oracle_ffi as O, pgx_ffi as P
G = O.GOLDEN + '/bidirectional_test/'
idx = P.Index(G + 'xy.ri', G + 'xy_bidirectional_compressed.tags')
ri = O.RIndex(G + 'xy.ri'); tg = O.Tags(G + 'xy_bidirectional_compressed.tags', O.TAGS_BYTECODE)
reads = [l for f in ('reads.txt', 'test_reads.txt') for l in open(G + f).read().split('\n') if l]
cat, offs = O.pack_reads(reads)
for ml in (5, 3):
    ref = O.find_mems_batch(ri, tg, cat, offs, ml, 1)
    print('ml', ml, 'ref mems', ref['mem_offsets'][-1], 'max runs', ref['tag_run_counts'].max() if len(ref['tag_run_counts']) else 0, flush=True)
    res = idx.find_mems(cat, offs, ml, 1, tags=False)
    print(' gpu mems ok', np.array_equal(res['mems'], ref['mems']), flush=True)
    rn, po, pos, nover = idx.tag_query_batch(ref['mems']['bwt_start'], ref['mems']['bwt_start'] + ref['mems']['size'].astype(np.uint64) - 1)
    print(' tag_query_batch ok', np.array_equal(rn, ref['tag_run_counts']), np.array_equal(pos, ref['positions']), flush=True)
    res = idx.find_mems(cat, offs, ml, 1, tags=True)
    print(' gpu tags ok', np.array_equal(res['positions'], ref['positions']), flush=True)
