"""Second, independent restatement of the index operators in numpy / pure Python (small cases).

TEST INFRASTRUCTURE.  Unlike oracle/gridgcn_oracle.c (a sequential simulation of schedule S0)
this file states the ORDER-INDEPENDENT form the HIP kernels implement:
  * bucket of a voxel   = its points in ascending id; item n >= P moves to slot r(n) if r(n) < P,
                          the largest n wins a slot;
  * centre slots        = voxels in order of first appearance; voxel t >= O moves to slot r(t),
                          the largest t wins;
  * query               = items flattened over the k^3 voxels; item g > P moves to slot r(g),
                          the largest g wins; total weight = sum over the final slots.
Agreement of the two restatements on random inputs is what justifies the kernels' design.
"""
import numpy as np

M32 = 0xFFFFFFFF


def xorwow_uniform(seed):
    seed &= 0xFFFFFFFFFFFFFFFF
    s0 = (seed & M32) ^ 0xaad26b49
    s1 = (seed >> 32) ^ 0xf7dcefdd
    t0 = (1099087573 * s0) & M32
    t1 = (2591861531 * s1) & M32
    d = (6615241 + t1 + t0) & M32
    v0 = (123456789 + t0) & M32
    v4 = (5783321 + t0) & M32
    t = v0 ^ (v0 >> 2)
    v4 = ((v4 ^ ((v4 << 4) & M32)) ^ (t ^ ((t << 1) & M32))) & M32
    d = (d + 362437) & M32
    x = (v4 + d) & M32
    return np.float32(np.float32(x) * np.float32(2.3283064e-10) + np.float32(1.1641532e-10))


def pick(seed, n):
    u = xorwow_uniform(seed)
    return int(np.ceil(np.float32(u * np.float32(n)))) - 1


def voxel_of(p, shift, vs, grid):
    c = []
    for j in range(3):
        q = np.float32(np.float32(p[j] + np.float32(shift[j])) / np.float32(vs[j]))
        f = np.floor(q)
        if not (f >= 0 and f < grid[j]):
            return -1, None
        c.append(int(f))
    return c[2] * grid[0] * grid[1] + c[1] * grid[0] + c[0], c


def _wrap32(x):
    x &= M32
    return x - (1 << 32) if x >= (1 << 31) else x


def gridify(data, actual_numpoints, *, max_p_grid, max_o_grid, kernel_size, stride=1, loc=0,
            coord_shift, voxel_size, grid_size, seed=0):
    data = np.asarray(data, np.float32)
    npnts = np.asarray(actual_numpoints).reshape(-1)
    B, N, _ = data.shape
    P, O, k = max_p_grid, max_o_grid, kernel_size
    k3 = k ** 3
    gx, gy, gz = grid_size
    nebidx = np.zeros((B, O, P), np.int32)
    nebmsk = np.zeros((B, O, P), np.float32)
    cent = np.ones((B, O, 4), np.float32)
    centmsk = np.zeros((B, O), np.float32)
    centnum = np.zeros((B, 1), np.int32)
    for b in range(B):
        members = {}
        for i in range(min(int(npnts[b]), N)):
            v, _ = voxel_of(data[b, i], coord_shift, voxel_size, grid_size)
            if v >= 0:
                members.setdefault(v, []).append(i)
        # buckets
        bucket = {}
        for v, ids in members.items():
            slots = list(ids[:P])
            for n in range(P, len(ids)):
                r = pick((b * N + ids[n]) + seed, n + 1)
                if r < P:
                    slots[r] = ids[n]      # ascending n: the last writer is the largest n
            bucket[v] = slots
        # centres
        order = sorted(members.keys(), key=lambda v: members[v][0])
        slot2vox = list(order[:O])
        for t in range(O, len(order)):
            r = pick((b * N + members[order[t]][0]) + 2 * seed, t + 1)
            if r < O:
                slot2vox[r] = order[t]
        cn = min(len(order), O)
        centnum[b, 0] = cn
        for o in range(cn):
            index = b * O + o
            v = slot2vox[o]
            c2 = v // (gx * gy)
            c1 = (v - c2 * gx * gy) // gx
            c0 = v - c2 * gx * gy - c1 * gx
            items = []
            for nei in range(k3):
                d = nei // (k * k) - (k - 1) // 2 + c2
                h = (nei % (k * k)) // k - (k - 1) // 2 + c1
                w = nei % k - (k - 1) // 2 + c0
                if 0 <= d < gz and 0 <= h < gy and 0 <= w < gx:
                    items += bucket.get(d * gx * gy + h * gx + w, [])
            slots = list(items[:P])
            for g in range(P + 1, len(items) + 1):
                s32 = _wrap32(_wrap32(index * P) * k3 + g)
                u = xorwow_uniform(s32 & 0xFFFFFFFFFFFFFFFF)
                r = int(np.ceil(np.float32(u * np.float32(g)))) - 1
                if r < P:
                    slots[r] = items[g - 1]
            m = len(slots)
            nebidx[b, o, :m] = slots
            nebidx[b, o, m:] = slots[0]
            nebmsk[b, o, :m] = 1.0
            cent[b, o, 3] = np.float32(sum(int(data[b, s, 3]) for s in slots))
            centmsk[b, o] = 1.0
            if loc == 1:
                sx = sy = sz = sw = np.float32(0)
                for i in members[v]:
                    x, y, z, wt = data[b, i]
                    sx = np.float32(sx + np.float32(x * wt))
                    sy = np.float32(sy + np.float32(y * wt))
                    sz = np.float32(sz + np.float32(z * wt))
                    sw = np.float32(sw + wt)
                cent[b, o, 0] = sx / sw
                cent[b, o, 1] = sy / sw
                cent[b, o, 2] = sz / sw
    return nebidx, nebmsk, cent, centmsk, centnum


def gridify_up(downdata, updata, down_np, up_np, *, max_p_grid, max_o_grid, kernel_size,
               coord_shift, voxel_size, grid_size, seed=0):
    down = np.asarray(downdata, np.float32)
    up = np.asarray(updata, np.float32)
    dnp = np.asarray(down_np).reshape(-1)
    unp = np.asarray(up_np).reshape(-1)
    B, Nd, _ = down.shape
    P, O, k = max_p_grid, max_o_grid, kernel_size
    k3 = k ** 3
    hk = (k - 1) // 2
    gx, gy, gz = grid_size
    nebidx = np.zeros((B, O, P), np.int32)
    nebmsk = np.zeros((B, O, P), np.float32)
    for b in range(B):
        own = {}
        for i in range(min(int(dnp[b]), Nd)):
            v, c = voxel_of(down[b, i], coord_shift, voxel_size, grid_size)
            if v >= 0:
                own[i] = c
        for o in range(O):
            if not o < unp[b]:
                continue
            vq, cq = voxel_of(up[b, o], coord_shift, voxel_size, grid_size)
            if vq < 0:
                continue
            cands = []   # (id, nei of the scatter thread)
            for i, c in own.items():
                dz, dy, dx = cq[2] - c[2], cq[1] - c[1], cq[0] - c[0]
                if max(abs(dz), abs(dy), abs(dx)) <= hk:
                    nei = (dz + hk) * k * k + (dy + hk) * k + (dx + hk)
                    cands.append((i, nei))
            cands.sort()
            slots = [c[0] for c in cands[:P]]
            for n in range(P, len(cands)):
                i, nei = cands[n]
                r = pick(seed + (b * Nd + i) * k3 + nei, n + 1)
                if r < P:
                    slots[r] = i
            m = len(cands)
            for j in range(P):
                if j < m:
                    nebidx[b, o, j] = slots[j]
                    nebmsk[b, o, j] = 1.0
                else:
                    nebidx[b, o, j] = slots[0] if m > 0 else 0
    return nebidx, nebmsk
