"""gfx950 layout facts the kernels rely on, checked on exact integer data through the C ABI."""
import numpy as np
import pytest
import torch

from carel_vae_amd import _lib as L

pytestmark = pytest.mark.gpu


def _ints(rs, shape, lo=-3, hi=4):
    return rs.randint(lo, hi, size=shape).astype(np.float32)


def test_layout_selftest():
    lib = L.load()
    L.check(lib.carel_init(0), "carel_init")
    rs = np.random.RandomState(3)
    A16, B16 = _ints(rs, (16, 32)), _ints(rs, (32, 16))
    A32, B32 = _ints(rs, (32, 16)), _ints(rs, (16, 32))
    A2 = _ints(rs, (32, 32), -2, 3)
    TA, TB = _ints(rs, (128, 64)), _ints(rs, (128, 64))        # A [m][k], B^T [n][k]
    # TR[r][c] = 16*(r % 16) + (c % 16): < 256 so exact in bf16, identifies (row, column-in-block)
    TR = (16 * (np.arange(32)[:, None] % 16) + (np.arange(128)[None, :] % 16)).astype(np.float32)
    inp = np.zeros(40960, dtype=np.float32)
    inp[0:512] = A16.ravel(); inp[512:1024] = B16.ravel()
    inp[1024:1536] = A32.ravel(); inp[1536:2048] = B32.ravel()
    inp[2048:3072] = A2.ravel()
    inp[4096:12288] = TA.ravel(); inp[12288:20480] = TB.ravel()
    inp[20480:28672] = TA.T.copy().ravel(); inp[28672:36864] = TB.T.copy().ravel()
    inp[36864:40960] = TR.ravel()
    d_in = torch.from_numpy(inp).cuda().to(torch.bfloat16)
    d_out = torch.full((73728,), float("nan"), device="cuda", dtype=torch.float32)
    L.check(lib.carel_selftest_layouts(d_in.data_ptr(), d_out.data_ptr(), L.current_stream()), "selftest")
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    report = []

    def cmp(name, got, ref):
        bad = int((got != ref).sum())
        report.append(f"{name}: {'ok' if bad == 0 else str(bad) + ' mismatches'}")
        return bad

    fails = 0
    fails += cmp("mfma16 A/B/D maps", out[0:256].reshape(16, 16), A16 @ B16)
    X = A32 @ B32
    fails += cmp("mfma32 A/B/D maps", out[256:1280].reshape(32, 32), X)
    tr = out[1280:1536].reshape(64, 4)
    exp = np.zeros((64, 4), dtype=np.float32)
    for l in range(64):
        g, i = l >> 4, l & 15
        for e in range(4):
            exp[l, e] = TR[4 * g + e, 16 * g + i]
    fails += cmp("ds_read_b64_tr_b16", tr, exp)
    dma = out[2048:2560]
    expd = np.concatenate([TA.ravel()[((l ^ 5) * 8):((l ^ 5) * 8 + 8)] for l in range(64)])
    fails += cmp("global_load_lds lane-linear", dma, expd)
    fails += cmp("acc tile as next B operand", out[4096:5120].reshape(32, 32), A2 @ X)
    C = TA @ TB.T
    for c, nm in enumerate(("NT", "NN", "TN", "TT")):
        fails += cmp(f"tile 128x128x64 {nm}", out[8192 + c * 16384: 8192 + (c + 1) * 16384].reshape(128, 128), C)
    print("\n".join(report))
    if fails:
        np.set_printoptions(linewidth=250)
        print("tr dump (lane x 4):\n", tr[:32])
        print("tr expected:\n", exp[:32])
    assert fails == 0, report
