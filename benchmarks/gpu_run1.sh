set -e
export TMPDIR=/tmp
STEPS=30 timeout -k 10 120 python benchmarks/vae_c4.py 2>&1 | grep "VAE"
STEPS=30 LIK=zinb timeout -k 10 120 python benchmarks/vae_c4.py 2>&1 | grep "VAE"
rm -rf gpurun_out/prof_vae
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vae -o vm -- python3 benchmarks/vae_c4.py > gpurun_out/vae_prof.log 2>&1
grep "VAE" gpurun_out/vae_prof.log
