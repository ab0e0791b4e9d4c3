set -x
mkdir -p gpurun_out/parity_r3

python scripts/parity_at_scale.py --reads 1500000 --spectrum > gpurun_out/parity_r3/parity_at_scale_spectrum.json 2> gpurun_out/parity_r3/d.err || { tail -5 gpurun_out/parity_r3/d.err; }
python scripts/parity_at_scale.py --reads 1000000 --spectrum --length 150 > gpurun_out/parity_r3/parity_at_scale_spectrum_150bp.json 2> gpurun_out/parity_r3/e.err || { tail -5 gpurun_out/parity_r3/e.err; }
cat gpurun_out/parity_r3/parity_at_scale_spectrum*.json | cut -c1-300
