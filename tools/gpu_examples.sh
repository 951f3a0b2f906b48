#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/examples
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for ex in "laplace2d.py epochs=300" "euler_beam.py epochs=300" "allen_cahn_plain.py epochs=2 iters_per_epoch=50" "tfno_darcyflow.py epochs=2" "uno_darcyflow.py epochs=2" "sfno_swe.py epochs=2" "spinn_helmholtz3d.py epochs=2 iters_per_epoch=20" "allen_cahn_piratenet.py epochs=1 iters_per_epoch=20" "poiseuille_flow.py epochs=20" "ldc2d_steady.py epochs=2 iters_per_epoch=20"; do
  set -- $ex
  echo "== $ex"
  timeout 600 python examples/$@ > gpurun_out/examples/$1.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/examples/$1.log | tail -2 | cut -c1-220
done
