import os, sys, subprocess, tempfile, time
sys.path.insert(0, os.getcwd())
from mapdamage_amd import fasta, sam, synth
ref = synth.make_genome()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
batch = synth.parallel_batch("config3_batch", ref, n, seed=3003, workers=16)
tmp = tempfile.mkdtemp(prefix="mdx_dbg_")
path = os.path.join(tmp, "c3.bam")
sam.write_bam(path, batch, ref.names, ref.lengths, [{"ID": "rg1", "SM": "synthetic", "LB": "lib1"}], rg_of_record="rg1", workers=16)
fasta.write_fasta(os.path.join(tmp, "ref.fa"), ref)
for k, extra in enumerate(({}, {"MDX_NO_FAST_EXIT": "1"}, {"MDX_GBAM_NO_LOOKAHEAD": "1"}, {"MDX_NO_WARM": "1"}, {})):
    env = dict(os.environ, PYTHONFAULTHANDLER="1", MDX_INIT_TRACE="1", **extra)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-m", "mapdamage_amd", "-i", path, "-r", os.path.join(tmp, "ref.fa"), "-d", os.path.join(tmp, "o%d" % k), "--no-stats", "--log-level", "DEBUG"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print("run", k, extra, "rc", p.returncode, "wall %.3f" % (time.perf_counter() - t0))
    print(p.stdout.decode()[-3000:])
