import sys; sys.path.insert(0, "/root/repo")
import numpy as np, panoptikon_amd as pvs, oracle as orc
dim=32
rows=orc.synth_rows(1,0,10,dim); grp=np.array([5,5,5,9,9,9,9,5,9,5],np.int64)
for devices in ([0,0,0],[0,0,0,0,0]):
    ix=pvs.VectorIndex(pvs.F32,dim,devices=devices)
    ix.add(rows,group_ids=grp)
    ix.set_order_keys(np.arange(10,dtype=np.int64))
    q=orc.synth_rows(2,0,2,dim)
    print(ix.search(q,4,pvs.L2)[0].tolist())
    print(ix.search(q,20,pvs.COSINE)[2].tolist())
    print(ix.search_groups(q,5,pvs.L2,pvs.AGG_AVG))
    print(ix.search_filtered(q,3,np.array([1,0,1,0,1,0,1,0,1,0],np.uint8),pvs.L2)[0].tolist())
    print(ix.similar_to(np.array([0,1],np.int64),5,pvs.L2,pvs.AGG_MIN))
    print(pvs.rrf_search([dict(index=ix,query=q[0],metric=pvs.L2,agg=pvs.AGG_MIN,rrf_k=1,weight=1.0)],5))
    print(ix.score_batch(q,pvs.L2).shape, ix.stats().rows)
    ei,ed=orc.search(orc.F32,orc.L2,rows,q,4,ids=np.arange(10,dtype=np.int64)); print(ei.tolist())
    ix.close()
# empty multi index
ix=pvs.VectorIndex(pvs.F32,dim,devices=[0,0])
print(ix.search(q,4,pvs.L2)[2].tolist())
ix.close()
print("done")
