import sys
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo')
from ablate import timeit
for n in (256, 512, 1024, 2048, 3072, 4096, 8192, 16384):
    print(n, round(timeit(1, n=n, steps=200),1), 'us/step', flush=True)
