export TMPDIR=/tmp
R=gpurun_out/r3; mkdir -p $R
stats() { name=$1; shift; rm -rf /tmp/st_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -o x -- "$@" > /tmp/st_$name.log 2>&1
  f=$(find /tmp/st_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/${name}_kernel_stats.csv; }
cat > /tmp/ivf_only.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from esrecsys_amd.ivf import IVFIndex
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(1701)
N, D, nq = 1 << 20, 512, 8192
cent = torch.randn((4096, D), generator=g, device=dev); cent = cent / cent.norm(dim=1, keepdim=True)
c = cent[torch.randint(0, 4096, (N,), generator=g, device=dev)] + torch.randn((N, D), generator=g, device=dev) * (0.6 * D ** -0.5)
q = cent[torch.randint(0, 4096, (nq,), generator=g, device=dev)] + torch.randn((nq, D), generator=g, device=dev) * (0.6 * D ** -0.5)
idx = IVFIndex(c, 1024)
for _ in range(6):
    idx.search(q, 500, 32)
torch.cuda.synchronize()
PY
stats ivf_k500_nprobe32 python /tmp/ivf_only.py
stats sharded_world1_machinery_triplet env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 python bench.py --workload triplet --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
stats sharded_world1_machinery_glove env ESR_BENCH_SHARDED=1 ESR_SHARDED_WORLD1_DIRECT=0 ESR_SHARDED_UNIQUE=0 python bench.py --workload glove --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
ls $R | grep -c kernel_stats
