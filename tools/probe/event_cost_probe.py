"""What an event record between two kernels of one stream costs that stream (the two-stream backward records one per weight gradient):
100 back-to-back kernels of ~50 us; the same with an event recorded after each (and a side stream waiting for it and running a small kernel);
the same with, in addition, the main stream waiting for the side stream's event of the launch before (the overwrite guard)."""
import torch
x = torch.randn(64 * 1024 * 1024, device="cuda")          # 256 MB: mul_ ~ 100 us
y = torch.randn(1024 * 1024, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def run(mode, n=100):
    evs = [torch.cuda.Event() for _ in range(n)]
    done = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        if mode >= 2 and i > 0:
            main.wait_event(done[i - 1])
        x.mul_(1.0001)
        if mode >= 1:
            evs[i].record(main)
            side.wait_event(evs[i])
            with torch.cuda.stream(side):
                y.mul_(1.0001)
                done[i].record(side)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(2):
    for mode, name in ((0, "kernels back to back"), (1, "+ an event record after each, a side stream waits for it"), (2, "+ the main stream waits for the side stream's previous event")):
        run(mode, 20)
        print("%-70s %.1f us per kernel" % (name, run(mode)))
