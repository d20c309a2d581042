#!/usr/bin/env python3
"""Index-dump fixtures for SURVEY.md section 8(f)-1, made by the REFERENCE ITSELF (oracle/_ref):

    python tests/golden/make_mmi_golden.py        (build container only; after make_golden.py)

  adv_k12w5_300K.mmi.gz   `minimap2-coverage -k 12 -w 5 -I 300K -d <out> adv_all.fa.gz`  (several parts, index.c:390-426)
  adv_k19w10.mmi.gz       `... -k 19 -w 10 -d <out> adv_all.fa.gz`                        (one part, other k / w)
  adv_mmi_k12w5.table.gz  the reference mapping adv_sub.fq.gz against the first file  (`-Y -l 0 -q 160 -p 160`)
  adv_mmi_k19w10.table.gz the same against the second file with the DEFAULT -k/-w on the command line: the index's
                          k / w drive the mapping, the command line's size the counters (minimap2-coverage.c:419-422)
Fixtures are data (the reference's output files); mmi_cases.json records the argv.
"""
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "minimap2-coverage")


def sh(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)
    if r.returncode != 0:
        raise SystemExit("failed: %s\n%s" % (" ".join(cmd), r.stderr.decode()[-2000:]))
    return r.stdout


def gz_write(path, data):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)
    return hashlib.md5(data).hexdigest()


def main():
    if not os.path.exists(REF):
        raise SystemExit("oracle/_ref is not built (make -C oracle ref)")
    os.chdir(HERE)
    cases = []
    with tempfile.TemporaryDirectory() as d:
        for name, build, mapv in [
            ("adv_k12w5_300K", ["-k", "12", "-w", "5", "-I", "300K"], ["-Y", "-l", "0", "-q", "160", "-p", "160", "-t", "4"]),
            ("adv_k19w10", ["-k", "19", "-w", "10"], ["-Y", "-l", "0", "-q", "160", "-p", "160", "-t", "4"]),
        ]:
            mmi = os.path.join(d, name + ".mmi")
            sh([REF] + build + ["-d", mmi, "adv_all.fa.gz"])
            md5_i = gz_write(name + ".mmi.gz", open(mmi, "rb").read())
            table = sh([REF] + mapv + [mmi, "adv_sub.fq.gz"])
            tname = name.replace("adv_", "adv_mmi_").replace("_300K", "") + ".table.gz"
            md5_t = gz_write(tname, table)
            cases.append(dict(name=name, build_argv=build + ["-d", name + ".mmi", "adv_all.fa.gz"], index=name + ".mmi.gz", index_md5=md5_i,
                              map_argv=mapv + [name + ".mmi", "adv_sub.fq.gz"], expect=tname, md5=md5_t))
    with open("mmi_cases.json", "w") as f:
        json.dump(cases, f, indent=1)
    os.system("ls -la *.mmi.gz adv_mmi_*")


if __name__ == "__main__":
    main()
