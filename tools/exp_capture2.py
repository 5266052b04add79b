"""scratch: minimal nested-fork capture patterns"""
import sys, torch
dev = torch.device("cuda:0")
pat = sys.argv[1]
x = torch.zeros(1 << 16, device=dev); y = torch.zeros_like(x); z = torch.zeros_like(x)
A, Bs = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def body():
    main = torch.cuda.current_stream()
    if pat == "nested":              # main -> A -> B, B joins A, A joins main
        A.wait_stream(main)
        with torch.cuda.stream(A):
            x.add_(1)
            Bs.wait_stream(A)
            with torch.cuda.stream(Bs):
                y.add_(1)
            x.add_(1)
            A.wait_stream(Bs)
            x.add_(1)
        main.wait_stream(A)
    elif pat == "nested_multi":      # B waits on A several times
        A.wait_stream(main)
        with torch.cuda.stream(A):
            for _ in range(3):
                x.add_(1)
                Bs.wait_stream(A)
                with torch.cuda.stream(Bs):
                    y.add_(1)
            A.wait_stream(Bs)
            x.add_(1)
        main.wait_stream(A)
    elif pat == "flat_multi":        # B waits on main several times
        for _ in range(3):
            x.add_(1)
            Bs.wait_stream(main)
            with torch.cuda.stream(Bs):
                y.add_(1)
        main.wait_stream(Bs)
    elif pat == "nested_multi_joinmain":
        A.wait_stream(main)
        with torch.cuda.stream(A):
            for _ in range(3):
                x.add_(1)
                Bs.wait_stream(A)
                with torch.cuda.stream(Bs):
                    y.add_(1)
            x.add_(1)
        main.wait_stream(A)
        main.wait_stream(Bs)
    z.add_(1)
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    body()
g.replay(); torch.cuda.synchronize()
print("OK", pat, float(x[0]), float(y[0]), float(z[0]), flush=True)
