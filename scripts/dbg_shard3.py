import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests.test_gpu_sharded import _run
if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for steps in (1, 2, 3):
        a = _run(W, steps, "sharded"); b = _run(W, steps, "replicated")
        print("steps", steps, "losses a", a[0][1], "b", b[0][1])
        nd = 0
        for k in b[0][2]:
            d = np.abs(a[0][2][k].astype(np.float64) - b[0][2][k]).max()
            if d > 0:
                nd += 1
                if nd < 8 or "embedding_trans/" in k and "embedding" == k.split("/")[-1]:
                    bad = np.argwhere(a[0][2][k] != b[0][2][k])
                    print("   ", k[-60:], a[0][2][k].shape, "maxdiff %.3g" % d, "ndiff", len(bad), "first", bad[:3].tolist())
        print("   vars differing:", nd, "predict diff", np.abs(a[0][3] - b[0][3]).max())
