"""A second, independent restatement of the reference algorithm (test infrastructure only).

Pure Python, written function-by-function against the Julia sources with the reference's own names,
1-based indexing (index 0 of every array is unused) and loop structure -- deliberately NOT sharing any
code or data layout decisions with oracle/kde_oracle.c.  It is slow and only meant for tiny cases: the
tests require the C oracle to agree with it exactly (labels) / to 1e-13 (points), so a misreading of
the reference would have to be made twice, in two different shapes, to go unnoticed.

Citations are relative to /root/reference/src.
"""
import math

NO_CHILD = -1  # BallTree01.jl:5
EPS = 2.220446049250313e-16  # eps(Float64)


class BT:  # BallTree01.jl:10-28 + BallTreeDensity01.jl:11-24 (flattened)
    pass


def _idx(i, dims, k):  # (i-1)*dims + k, 1-based
    return (i - 1) * dims + k


# ---- BallTree01.jl -------------------------------------------------------------------------------------

def validIndex(bt, ind):  # :83
    return 0 < ind <= 2 * bt.num_points


def swapDensity(bd, i, j):  # swapBall! :109-138 + swapDensity! BallTreeDensity01.jl:112-139
    if i == j:
        return
    bd.weights[i], bd.weights[j] = bd.weights[j], bd.weights[i]
    bd.permutation[i], bd.permutation[j] = bd.permutation[j], bd.permutation[i]
    for k in range(1, bd.dims + 1):
        a, b = _idx(i, bd.dims, k), _idx(j, bd.dims, k)
        bd.centers[a], bd.centers[b] = bd.centers[b], bd.centers[a]
        bd.means[a], bd.means[b] = bd.means[b], bd.means[a]
        bd.bandwidth[a], bd.bandwidth[b] = bd.bandwidth[b], bd.bandwidth[a]


def most_spread_coord(bt, low, high):  # :142-173
    max_variance = 0
    max_dim = 1
    w = 1.0 / (high - low)
    for dimension in range(1, bt.dims + 1):
        mean = 0
        # (dims*(low-1) + dimension):dims:(dims*(high-1))  -- the last leaf is never reached
        point = bt.dims * (low - 1) + dimension
        stop = bt.dims * (high - 1)
        pts = []
        while point <= stop:
            pts.append(point)
            point += bt.dims
        for p in pts:
            mean = mean + w * bt.centers[p]
        variance = 0
        for p in pts:
            variance += (bt.centers[p] - mean) ** 2
        if variance > max_variance:
            max_variance = variance
            max_dim = dimension
    return max_dim


def select(bd, dimension, position, low, high):  # :223-242
    while low < high:
        r = (low + high) // 2
        swapDensity(bd, r, low)
        m = low
        for i in range(low, high + 1):
            if bd.centers[dimension + bd.dims * (i - 1)] - bd.centers[dimension + bd.dims * (low - 1)] < 0.0:
                m += 1
                swapDensity(bd, m, i)
        swapDensity(bd, low, m)
        if m <= position:
            low = m + 1
        if m >= position:
            high = m - 1


def calcStatsDensity(bd, root):  # calcStatsBall! :282-336 + calcStatsDensity! BallTreeDensity01.jl:141-187
    leftI, rightI = bd.left_child[root], bd.right_child[root]
    if not validIndex(bd, leftI) or not validIndex(bd, rightI):
        return
    D = bd.dims
    for d in range(1, D + 1):
        a = bd.centers[_idx(leftI, D, d)] + bd.ranges[_idx(leftI, D, d)]
        b = bd.centers[_idx(rightI, D, d)] + bd.ranges[_idx(rightI, D, d)]
        maxi = a if a > b else b
        c = bd.centers[_idx(leftI, D, d)] - bd.ranges[_idx(leftI, D, d)]
        c2 = bd.centers[_idx(rightI, D, d)] - bd.ranges[_idx(rightI, D, d)]
        mini = c if c < c2 else c2
        halfspan = (maxi - mini) / 2.0
        bd.ranges[_idx(root, D, d)] = halfspan
        bd.centers[_idx(root, D, d)] = mini + halfspan
    if leftI != rightI:
        bd.weights[root] = bd.weights[leftI] + bd.weights[rightI]
    else:
        bd.weights[root] = bd.weights[leftI]
    Ni, NiL, NiR = D * (root - 1), D * (leftI - 1), D * (rightI - 1)
    wtL, wtR = bd.weights[leftI], bd.weights[rightI]
    wtT = wtL + wtR + EPS
    wtL /= wtT
    wtR /= wtT
    for k in range(1, D + 1):
        bd.means[Ni + k] = wtL * bd.means[NiL + k] + wtR * bd.means[NiR + k]
        bd.bandwidth[Ni + k] = (wtL * (bd.bandwidth[NiL + k] + bd.means[NiL + k] * bd.means[NiL + k]) +
                                wtR * (bd.bandwidth[NiR + k] + bd.means[NiR + k] * bd.means[NiR + k]) -
                                bd.means[Ni + k] * bd.means[Ni + k])


def buildBall(bd, low, high, root):  # :342-411
    if low == high:
        bd.lowest_leaf[root] = low
        bd.highest_leaf[root] = high
        bd.left_child[root] = low
        bd.right_child[root] = high
        calcStatsDensity(bd, root)
        bd.right_child[root] = NO_CHILD
        return
    coord = most_spread_coord(bd, low, high)
    split = (low + high) // 2
    select(bd, coord, split, low, high)
    if split <= low:
        left = low
    else:
        left = bd.next
        bd.next += 1
    if split + 1 >= high:
        right = high
    else:
        right = bd.next
        bd.next += 1
    bd.lowest_leaf[root] = low
    bd.highest_leaf[root] = high
    bd.left_child[root] = left
    bd.right_child[root] = right
    if left != low:
        buildBall(bd, low, split, left)
    if right != high:
        buildBall(bd, split + 1, high, right)
    calcStatsDensity(bd, root)


def kde(points, ks, weights=None):
    """kde!(points, ks, weights) KDE01.jl:34-57 -> makeBallTreeDensity BallTreeDensity01.jl:192-231.
    points: list of N points, each a list of D floats.  Returns the flattened density (1-based lists)."""
    N, D = len(points), len(points[0])
    if len(ks) == 1:
        ks = list(ks) * D
    ks = [k * k for k in ks]
    if weights is None:
        weights = [1.0] * N
    sw = 0.0
    for x in weights:
        sw += x
    weights = [x / sw for x in weights]
    bd = BT()
    bd.dims, bd.num_points = D, N
    z = lambda n, v=0.0: [None] + [v] * n  # noqa: E731  (1-based array)
    bd.centers, bd.ranges, bd.means, bd.bandwidth = z(2 * N * D), z(2 * N * D), z(2 * N * D), z(2 * N * D)
    bd.weights = z(2 * N)
    bd.left_child, bd.right_child = z(2 * N, 1), z(2 * N, 1)
    bd.lowest_leaf, bd.highest_leaf = z(2 * N, 1), z(2 * N, 1)
    bd.permutation = z(2 * N, 0)
    for i in range(N):
        for k in range(D):
            bd.centers[N * D + i * D + k + 1] = points[i][k]
            bd.means[N * D + i * D + k + 1] = points[i][k]
            bd.bandwidth[N * D + i * D + k + 1] = ks[k]
        bd.weights[N + i + 1] = weights[i]
    # buildTree! BallTree01.jl:415-434
    i = N
    for j in range(1, N + 1):
        for k in range(1, D + 1):
            bd.ranges[i * D + k] = 0
        i += 1
        bd.lowest_leaf[i] = i
        bd.highest_leaf[i] = i
        bd.left_child[i] = i
        bd.right_child[i] = NO_CHILD
        bd.permutation[i] = j
    bd.next = 2
    buildBall(bd, N + 1, 2 * N, 1)
    return bd


# ---- MSGibbs01.jl ---------------------------------------------------------------------------------------

class Glb:
    pass


def mean_(bd, i, k):
    return bd.means[(i - 1) * bd.dims + k]


def bw_(bd, i, k):
    return bd.bandwidth[(i - 1) * bd.dims + k]


def updateGlbParticlesVariance(glb, j):  # :89-115
    for dim in range(1, glb.Ndim + 1):
        if not glb.partialDimMask[j][dim]:
            glb.particles[dim][j] = 0.0
            glb.variance[dim][j] = 0.0
        else:
            glb.particles[dim][j] = mean_(glb.trees[j], glb.ind[j], dim)
            glb.variance[dim][j] = bw_(glb.trees[j], glb.ind[j], dim)


def calcIndices(glb):  # :123-130
    for j in range(1, glb.Ndens + 1):
        updateGlbParticlesVariance(glb, j)


# ---- the operator tuples addop / diffop / getMu / getLambda of src/MSGibbs01.jl:650-653, per dimension, as CALLABLES (the
# reference's own interface: its callers bring on-manifold functions).  Defaults = the reference's Euclidean ones; the
# circular set is this repo's stated semantic (include/kdehip.h "manifolds"), which the enumerated C / HIP paths must match.
def getEuclidLambda(lambdas):  # :141
    lam = 0.0
    for v in lambdas:
        lam += v
    return lam


def getEuclidMu(mus, lambdas, scale=1.0):  # :152-161
    lambdamu = 0.0
    for z in range(len(mus)):
        lambdamu += mus[z] * lambdas[z]
    return scale * lambdamu


TWO_PI = 6.283185307179586476925286766559


def wrapRad(t):
    return t - TWO_PI * math.floor((t + math.pi) / TWO_PI)


def circ_diff(a, b):
    return wrapRad(a - b)


def circ_add(a, b):
    return wrapRad(a + b)


getCircLambda = getEuclidLambda


def getCircMu(mus, lambdas, scale=1.0):
    ref = 0.0
    for z in range(len(mus)):
        if lambdas[z] > 0.0:
            ref = mus[z]
            break
    acc = 0.0
    for z in range(len(mus)):
        acc += lambdas[z] * circ_diff(mus[z], ref)
    return circ_add(ref, scale * acc)


EUCLID_OPS = (lambda a, b: a + b, lambda a, b: a - b, getEuclidMu, getEuclidLambda)
CIRCULAR_OPS = (circ_add, circ_diff, getCircMu, getCircLambda)


def gaussianProductMeanCov(glb, dim, skip):  # :176-216, returns (destMu, destCov)
    checkpartials = [None] + [glb.partialDimMask[j][dim] for j in range(1, glb.Ndens + 1)]
    if skip > 0:
        checkpartials[skip] = False
    if not any(checkpartials[1:]):
        return 0.0, 0.0
    calclambdas, calcmu = [None] * (glb.Ndens + 1), [None] * (glb.Ndens + 1)
    for j in range(1, glb.Ndens + 1):
        if j != skip and glb.partialDimMask[j][dim]:
            calclambdas[j] = 1.0 / glb.variance[dim][j]
            calcmu[j] = glb.particles[dim][j]
        else:
            calclambdas[j] = 0.0
            calcmu[j] = 0.0
    destCov = glb.getLambda[dim](calclambdas[1:])  # :210
    destCov = 1.0 / destCov                         # :211
    return glb.getMu[dim](calcmu[1:], calclambdas[1:], destCov), destCov  # :213


def makeFasterSampleIndex(j, glb, muValue, covValue, offset, doCalmost):  # :250-328
    pT = 0.0
    zz = glb.levelList[j][1]
    dimmask = [None] + [False] * glb.Ndim
    for jj in range(1, glb.Ndens + 1):
        if jj == j:
            continue
        for d in range(1, glb.Ndim + 1):
            dimmask[d] = dimmask[d] or glb.partialDimMask[jj][d]
    tree = glb.trees[j]
    for z in range(1, glb.dNpts[j] + 1):
        glb.p[z] = 0.0
        for i in range(1, glb.Ndim + 1):
            if not glb.partialDimMask[j][i] or not dimmask[i]:
                continue
            tmpC = bw_(tree, zz, i)
            if doCalmost:
                tmpC += covValue[i]
            tmpM = glb.diffop[i](mean_(tree, zz, i), muValue[i + offset])  # :290
            try:
                distr = (tmpM * tmpM) / tmpC
            except ZeroDivisionError:
                distr = float("nan") if tmpM == 0.0 else float("inf")
            if not math.isnan(distr):
                glb.p[z] += distr
                glb.p[z] += math.log(tmpC) if tmpC > 0 else (float("-inf") if tmpC == 0 else float("nan"))
        try:
            e = math.exp(-0.5 * glb.p[z])
        except OverflowError:
            e = float("inf")
        glb.p[z] = e * tree.weights[zz]
        if math.isnan(glb.p[z]):
            glb.p[z] = 0.0
        pT += glb.p[z]
        if z < glb.dNpts[j]:
            zz = glb.levelList[j][z + 1]
    if pT < 1e-99:
        w = tree.weights[zz]
        pT = 0.0
        for z in range(1, glb.dNpts[j] + 1):
            glb.p[z] = w
            pT += w
    for z in range(1, glb.dNpts[j] + 1):
        glb.p[z] /= pT
    for z in range(2, glb.dNpts[j] + 1):
        glb.p[z] += glb.p[z - 1]


def selectLabelOnLevel(glb, j):  # :330-351
    dNp = glb.dNpts[j]
    z = 1
    zz = glb.levelList[j][z]
    while z <= dNp - 1:
        if glb.randU[glb.ruptr] <= glb.p[z]:  # ruptr is read BEFORE it is incremented (1-based array)
            break
        z += 1
        if z <= dNp:
            zz = glb.levelList[j][z]
    glb.ind[j] = zz
    glb.ruptr += 1


def sampleIndices(X, glb, offset):  # :364-385
    for j in range(1, glb.Ndens + 1):
        makeFasterSampleIndex(j, glb, X, None, offset, False)
        selectLabelOnLevel(glb, j)
    calcIndices(glb)


def sampleIndex(j, glb):  # :404-429
    for i in range(1, glb.Ndim + 1):
        glb.Malmost[i], glb.Calmost[i] = gaussianProductMeanCov(glb, i, j)
    makeFasterSampleIndex(j, glb, glb.Malmost, glb.Calmost, 0, True)
    selectLabelOnLevel(glb, j)
    updateGlbParticlesVariance(glb, j)


def samplePoint(X, glb, idx, addEntropy=True):  # :440-463
    for dim in range(1, glb.Ndim + 1):
        mn, vn = gaussianProductMeanCov(glb, dim, -1)
        glb.rnptr += 1
        if addEntropy:
            X[dim + idx] = glb.addop[dim](mn, math.sqrt(vn) * glb.randN[glb.rnptr])  # :456
        else:
            X[dim + idx] = mn


def levelInit(glb):  # :467-475
    for j in range(1, glb.Ndens + 1):
        glb.dNpts[j] = 1
        glb.levelList[j][1] = 1


def initIndices(glb):  # :477-497
    for j in range(1, glb.Ndens + 1):
        dNp = glb.dNpts[j]
        zz = glb.levelList[j][1]
        z = 1
        while z <= dNp:
            glb.p[z] = glb.trees[j].weights[zz]
            z += 1
            if z <= dNp:
                zz = glb.levelList[j][z]
        for z in range(2, dNp + 1):
            glb.p[z] += glb.p[z - 1]
        selectLabelOnLevel(glb, j)


def levelDown(glb):  # :500-523
    for j in range(1, glb.Ndens + 1):
        tree = glb.trees[j]
        z = 1
        for y in range(1, glb.dNpts[j] + 1):
            node = glb.levelList[j][y]
            if validIndex(tree, tree.left_child[node]):
                glb.levelListNew[j][z] = tree.left_child[node]
                z += 1
            if validIndex(tree, tree.right_child[node]):
                glb.levelListNew[j][z] = tree.right_child[node]
                z += 1
            if glb.ind[j] == node:
                glb.ind[j] = glb.levelListNew[j][z - 1]
        glb.dNpts[j] = z - 1
    glb.levelList, glb.levelListNew = glb.levelListNew, glb.levelList


def prodAppxMSGibbsS(trees0, Np, Niter, randU0, randN0, addEntropy=True, partialDimMask0=None, ops=None):
    """gibbs1 :527-629 as called by prodAppxMSGibbsS :645-703.  trees0: list of densities from kde();
    randU0 / randN0: 0-based Python lists.  Returns (points[d][s], indices[j][s]) 0-based nested lists."""
    glb = Glb()
    Ndens = len(trees0)
    glb.Ndens = Ndens
    glb.trees = [None] + list(trees0)
    glb.Ndim = max(t.dims for t in trees0)
    # ops: per dimension an (addop, diffop, getMu, getLambda) tuple -- EUCLID_OPS (default) or CIRCULAR_OPS (:650-653, 672-675)
    ops = [EUCLID_OPS] * glb.Ndim if ops is None else list(ops)
    glb.addop = [None] + [o[0] for o in ops]
    glb.diffop = [None] + [o[1] for o in ops]
    glb.getMu = [None] + [o[2] for o in ops]
    glb.getLambda = [None] + [o[3] for o in ops]
    glb.randU = [None] + list(randU0)
    glb.randN = [None] + list(randN0)
    if partialDimMask0 is None:
        partialDimMask0 = [[True] * glb.Ndim for _ in range(Ndens)]
    glb.partialDimMask = [None] + [[None] + [bool(b) for b in m] for m in partialDimMask0]
    maxNp = max(t.num_points for t in trees0)
    glb.ind = [None] + [1] * Ndens
    glb.p = [None] + [0.0] * maxNp
    glb.Malmost = [None] + [0.0] * glb.Ndim
    glb.Calmost = [None] + [0.0] * glb.Ndim
    glb.Nlevels = int(math.floor(math.log(maxNp) / math.log(2) + 1))
    glb.particles = [None] + [[None] + [0.0] * Ndens for _ in range(glb.Ndim)]
    glb.variance = [None] + [[None] + [0.0] * Ndens for _ in range(glb.Ndim)]
    glb.dNpts = [None] + [0] * Ndens
    glb.levelList = [None] + [[None] + [1] * maxNp for _ in range(Ndens)]
    glb.levelListNew = [None] + [[None] + [1] * maxNp for _ in range(Ndens)]
    glb.ruptr = 0
    glb.rnptr = 0
    newPoints = [None] + [0.0] * (glb.Ndim * Np)
    newIndices = [[0] * Np for _ in range(Ndens)]
    for s in range(1, Np + 1):
        frm = (s - 1) * glb.Ndim
        levelInit(glb)
        initIndices(glb)
        calcIndices(glb)
        for _l in range(1, glb.Nlevels + 1):
            samplePoint(newPoints, glb, frm)
            levelDown(glb)
            sampleIndices(newPoints, glb, frm)
            for _i in range(1, Niter + 1):
                for j in range(1, Ndens + 1):
                    sampleIndex(j, glb)
        for j in range(1, Ndens + 1):
            newIndices[j - 1][s - 1] = glb.trees[j].permutation[glb.ind[j]] + 1
        samplePoint(newPoints, glb, frm, addEntropy)
    pts = [[newPoints[(s - 1) * glb.Ndim + d] for s in range(1, Np + 1)] for d in range(1, glb.Ndim + 1)]
    return pts, newIndices
