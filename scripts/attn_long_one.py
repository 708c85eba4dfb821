import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.attn_long_micro import run
if __name__ == "__main__":
    run(4096, 200, 200, 4, 80, 1, 1, iters=3)
