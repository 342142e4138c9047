def inverse(p):
    """e3nn.math.perm.inverse: q with q[p[i]] = i."""
    q = [0] * len(p)
    for i, j in enumerate(p):
        q[j] = i
    return tuple(q)
