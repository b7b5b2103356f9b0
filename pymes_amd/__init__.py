"""pymes_amd — MI355X-native CCSD/DCSD amplitude-update engine with the interfaces of
nickirk/pymes' solver path (pymes.solver.{ccsd,ccd,mp2}, pymes.mixer.diis,
pymes.integral.partition, pymes.util.fcidump, pymes.mean_field.hf).

The compute path is hand-written HIP for gfx950 behind the C-ABI of include/pymes_amd.h
(pymes_amd/lib/libpymes_amd.so); there is no CPU fallback."""
__version__ = "0.1.0"
