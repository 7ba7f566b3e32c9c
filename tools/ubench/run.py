import ctypes, os, sys
import torch
torch.zeros(1, device="cuda")
name = {"2": "libvalu_rates2.so", "clock": "libclock_probe.so"}.get(sys.argv[1] if len(sys.argv) > 1 else "", "libvalu_rates.so")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), name))
lib.ubench_main()
