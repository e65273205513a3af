import torch, time, sys
x = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
print("holding", flush=True)
time.sleep(float(sys.argv[1]))
