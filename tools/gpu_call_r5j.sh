#!/bin/bash
# round 5, GPU call J: the dense-SDF leg at the reference's own mesh-extraction size (scripts/extract_mesh.py: resolution 1024 -> 2^30 lattice
# points), and the dispatcher census of one training step of configs 2 and 5 on the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
timeout 300 python - > $O/dense_1024.json 2> $O/dense_1024.err <<'PY'
import json, torch, bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
model = bench.build_model(dev)
out = bench.dense_sdf_leg(model, dev, resolution=(1024, 1024, 1024), reps=2)
from sdfstudio_amd import build as _b
out["library_digest"] = _b.built_digest()
print(json.dumps(out))
PY
tail -c 300 $O/dense_1024.err; python -c "
import json; d=json.loads(open('gpurun_out/r5j/dense_1024.json').read().strip().splitlines()[-1]); print('dense 1024^3', d['ms'], d['value'], d['roofline']['frac'], d['kernels_ms'], d['launches'])"
timeout 200 python tools/aten_census.py 2 > $O/aten_census_cfg2.txt 2>&1; tail -4 $O/aten_census_cfg2.txt | cut -c1-200
timeout 300 python tools/aten_census.py 5 200000 > $O/aten_census_cfg5.txt 2>&1; tail -4 $O/aten_census_cfg5.txt | cut -c1-200
