import os, time, tempfile, shutil
for base in ("/dev/shm", "/tmp", os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"):
    try:
        td = tempfile.mkdtemp(dir=base)
    except Exception as e:
        print(base, e); continue
    t = time.perf_counter()
    for i in range(100):
        fd = os.open(os.path.join(td, "Group_%d_ACGTACGTACGTACGTACGTACGTACGT.fa" % i), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666); os.close(fd)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    for i in range(100):
        fd = os.open(os.path.join(td, "Group_%d_ACGTACGTACGTACGTACGTACGTACGT.fa" % i), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666); os.close(fd)
    dt2 = time.perf_counter() - t
    print("%s: create 100 files %.4f s, reopen+truncate %.4f s; fs: %s" % (base, dt, dt2, os.popen("stat -f -c %T " + td).read().strip()))
    shutil.rmtree(td)
