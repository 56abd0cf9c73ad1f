"""Mirrors of the library's routing rules that tests assert against (which kernel family answers a search)."""


def lists_fit(dtype, dim, k, batch):
    """plan_capw (csrc/pvs_direct.hip): does a batch of `batch` queries for pages of k rows share ONE launch of the exact search?"""
    esz = {"i8": 1, "f16": 2, "f32": 4}[dtype]
    stride = (dim * esz + 255) // 256 * 256
    w = 1 if batch <= 1 else 2 if batch <= 2 else 4 if batch <= 4 else 8
    if batch > 8 or k > 256 or (dtype != "i8" and w > 4):
        return False
    qbytes = w * (stride if esz == 1 else stride // esz * 4)
    if qbytes + 1024 > 30 * 1024:
        return False
    kp = 16
    while kp < k:
        kp *= 2
    capw = min(max(2 * kp, 128), (30 * 1024 - qbytes) // (32 * w) // 32 * 32, 512)
    return capw >= k + 64 and w * k <= 1024
