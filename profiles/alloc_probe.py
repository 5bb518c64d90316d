import time, torch
dev = torch.device("cuda:0")
total = 1464361
def T(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print("empty 96", T(lambda: torch.empty((total, 96), device=dev)))
print("empty 16", T(lambda: torch.empty((total, 16), device=dev)))
print("empty 96 + fill", T(lambda: torch.empty((total, 96), device=dev).fill_(1.0)))
print(torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
