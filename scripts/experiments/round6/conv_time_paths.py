import sys, os, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts/experiments/round6"))
import torch
from locality import B, M, base_args
dev = torch.device("cuda", 0)
for name, mat, kw in (("webbase 0.3 uniform", M.webbase_like(), {}), ("webbase 0.3 powerlaw", M.webbase_like(far="powerlaw"), {}),
                      ("webbase 0.3 powerlaw slabs=0", M.webbase_like(far="powerlaw"), dict(slabs="0")),
                      ("webbase 0.9", M.webbase_like(band=0.9), {})):
    p = B.Problem(mat, name, "f64", base_args(**kw), dev, 14)
    i = p.info
    print(f"{name:32s} asCSR5 {p.convert_ms:7.3f} ms (first {p.convert_first_ms:7.3f})  slabs={i.column_slabs} hot={i.slab_hot} cover={i.slab_hot_cover_pct}% t_slab={i.t_slab_ms:.3f} ms")
    p.close()
