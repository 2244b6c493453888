#!/usr/bin/env python3
"""LDS bank conflicts of the step kernels' random row reads, simulated from the real block tables.

    python3 tools/lds_conflict_sim.py dump      (on the GPU box: C2 plan -> gpurun_out/tab_dump.npz, 600 blocks of the loss table)
    python3 tools/lds_conflict_sim.py           (anywhere: the simulation on the dump)

A `ds_read_b128` of a wavefront is served in four groups of 16 lanes (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27}, {4-11,16-19,28-31}
and the same + 32); a group takes as many LDS cycles as the fullest of the sixteen 16-byte slots of the 256-byte bank row has DISTINCT
addresses.  The staged rows sit at position = rank of the row in the block's ascending list, so the slot is position mod 16.  The
script counts the cycles per group over every (slot of the table, wavefront, lane group) of the sampled blocks as placed, and after a
greedy re-placement that gives rows read together different residues mod 16 -- once with every row free to move, once with the block's
own 256 rows kept where they are (what every kernel assumes today).  Round 5, C2: 2.65 cycles per group as placed (the measured
conflict share 0.62 of the float64 kernel), 1.92 with every row free, 2.33 with the own rows fixed."""
import sys
def dump():
    sys.path.insert(0,'/root/repo')
    from depth_correction_amd.dataset import RoomBoxDataset
    from depth_correction_amd.pipeline import build_sequence
    dev=torch.device('cuda:0')
    ds = RoomBoxDataset(n_pts=200_000, n_poses=10, seed_base=1000, dtype=np.float32)
    scans = [np.stack([c[f] for f in 'xyz'], axis=1) for c, _ in ds]
    poses = np.stack([p for _, p in ds])
    plan, info = build_sequence(scans, poses, k=10, dtype=torch.float32, device=dev)
    ft = plan.fwd_table_loss or plan.fwd_table
    nb = 600
    sp = ft.slot_ptr.cpu().numpy()
    loc = ft.loc.cpu().numpy().reshape(-1, 256)
    bp = ft.blk_ptr.cpu().numpy()
    ob = ft.own_base.cpu().numpy() if ft.own_base is not None else None
    skip = plan.blk_skip.cpu().numpy() if plan.blk_skip is not None else None
    mask = plan.mask.cpu().numpy() if getattr(plan,'mask',None) is not None else None
    sel = np.arange(2000, 2000+nb)
    np.savez_compressed('/root/repo/gpurun_out/tab_dump.npz', slot_ptr=sp[sel[0]:sel[-1]+2], loc=loc[sp[sel[0]]:sp[sel[-1]+1]], blk_ptr=bp[sel[0]:sel[-1]+2], own_base=ob[sel] if ob is not None else np.zeros(0), skip=skip[sel] if skip is not None else np.zeros(0))
    print('ok', loc.shape, ft.max_rows)


def simulate():
    import numpy as np
    d=np.load(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'gpurun_out', 'tab_dump.npz'))
    sp=d['slot_ptr']; loc=d['loc']; bp=d['blk_ptr']; ob=d['own_base']; skip=d['skip']
    G=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
    G=G+[[l+32 for l in g] for g in G]
    def cycles(pos_rows):   # pos_rows: [10,256] positions (or -1)
        tot=0; ideal=0
        for q in range(pos_rows.shape[0]):
            for w in range(4):
                lanes=pos_rows[q, w*64:(w+1)*64]
                if (lanes<0).all(): continue
                for g in G:
                    p=lanes[g]; p=p[p>=0]
                    if len(p)==0: continue
                    u=np.unique(p)
                    cnt=np.bincount(u%16, minlength=16)
                    tot+=cnt.max(); ideal+=1
        return tot, ideal
    T=I=0; T2=0; T3=0
    rng=np.random.default_rng(0)
    nb=len(sp)-1
    for b in range(0,nb,6):
        if len(skip) and skip[b]: continue
        rows=loc[sp[b]-sp[0]:sp[b+1]-sp[0]].astype(np.int64)
        pos=np.where(rows==0xFFFF, -1, rows>>4)
        t,i=cycles(pos); T+=t; I+=i
        # greedy recolouring of ALL rows: new position = colour + 16*m ; colour chosen to minimise conflicts
        n=int(bp[b+1]-bp[b])
        # build sets
        sets=[]
        for q in range(pos.shape[0]):
            for w in range(4):
                lanes=pos[q,w*64:(w+1)*64]
                for g in G:
                    p=np.unique(lanes[g][lanes[g]>=0])
                    if len(p)>1: sets.append(p)
        member=[[] for _ in range(n)]
        for si,s in enumerate(sets):
            for r in s: member[r].append(si)
        used=np.zeros((len(sets),16),dtype=np.int32)
        colour=np.full(n,-1); cap=np.zeros(16,dtype=np.int32); capmax=(n+15)//16+2
        order=np.argsort([-len(m) for m in member])
        for r in order:
            cost=used[member[r]].sum(axis=0) if member[r] else np.zeros(16)
            cost=cost+ (cap>=capmax)*1000
            c=int(np.argmin(cost+rng.random(16)*0.01)); colour[r]=c; cap[c]+=1
            for si in member[r]: used[si,c]+=1
        # assign positions
        newpos=np.zeros(n,dtype=np.int64); nxt=np.zeros(16,dtype=np.int64)
        for r in range(n):
            newpos[r]=colour[r]+16*nxt[colour[r]]; nxt[colour[r]]+=1
        pos2=np.where(pos>=0, newpos[np.clip(pos,0,n-1)], -1)
        t2,_=cycles(pos2); T2+=t2
    print('b128 group-cycles: current %d ideal %d ratio %.3f ; recoloured(all rows free) %d ratio %.3f' % (T,I,T/I,T2,T2/I))
    
    # ---- variant: own rows keep their consecutive positions, the others are permuted among the remaining positions (dense)
    T=I=T2=0
    for b in range(0,nb,6):
        if len(skip) and skip[b]: continue
        rows=loc[sp[b]-sp[0]:sp[b+1]-sp[0]].astype(np.int64)
        pos=np.where(rows==0xFFFF, -1, rows>>4)
        n=int(bp[b+1]-bp[b]); own=int(ob[b])
        t,i=cycles(pos); T+=t; I+=i
        sets=[]
        for q in range(pos.shape[0]):
            for w in range(4):
                lanes=pos[q,w*64:(w+1)*64]
                for g in G:
                    p=np.unique(lanes[g][lanes[g]>=0])
                    if len(p)>1: sets.append(p)
        member=[[] for _ in range(n)]
        for si,s in enumerate(sets):
            for r in s: member[r].append(si)
        used=np.zeros((len(sets),16),dtype=np.int32)
        is_own=np.zeros(n,bool); is_own[own:own+256]=True
        for r in np.nonzero(is_own)[0]:
            for si in member[r]: used[si, r%16]+=1
        free=[p for p in range(n) if not is_own[p]]
        avail=np.bincount(np.array(free)%16, minlength=16) if free else np.zeros(16,int)
        others=[r for r in range(n) if not is_own[r]]
        others.sort(key=lambda r:-len(member[r]))
        colour={}
        for r in others:
            cost=(used[member[r]].sum(axis=0) if member[r] else np.zeros(16)) + (avail<=0)*1000
            c=int(np.argmin(cost+rng.random(16)*0.01)); colour[r]=c; avail[c]-=1
            for si in member[r]: used[si,c]+=1
        byc={c:[p for p in free if p%16==c] for c in range(16)}
        newpos=np.arange(n)
        for r in others: newpos[r]=byc[colour[r]].pop()
        pos2=np.where(pos>=0, newpos[np.clip(pos,0,n-1)], -1)
        t2,_=cycles(pos2); T2+=t2
    print('own rows fixed: current ratio %.3f ; others recoloured %.3f' % (T/I, T2/I))


if __name__ == '__main__':
    dump() if sys.argv[1:] == ['dump'] else simulate()
