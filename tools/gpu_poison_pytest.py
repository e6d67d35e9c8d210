"""diagnostic: run a pytest selection in a process whose cached GPU memory is pre-filled with NaN, so that any kernel
reading memory it (or a memset) never wrote shows up deterministically instead of depending on what the previous tenant
of the box left behind.  usage: python tools/gpu_poison_pytest.py <pytest args>"""
import sys, torch, pytest
big = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(12)]        # 12 x 1 GiB blocks
small = [torch.full((1 << 17,), float("nan"), device="cuda") for _ in range(1500)]    # 1500 x 512 KiB (small pool)
tiny = [torch.full((256,), float("nan"), device="cuda") for _ in range(4000)]
del big, small, tiny
torch.cuda.synchronize()
sys.exit(pytest.main(sys.argv[1:]))
