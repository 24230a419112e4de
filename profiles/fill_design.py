"""Regenerates the tables of DESIGN.md section 0 from the committed summaries: python profiles/fill_design.py r04
(between the <!-- NAME --> ... <!-- /NAME --> markers; the first run replaces a bare <!-- NAME --> marker)."""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
rnd = sys.argv[1] if len(sys.argv) > 1 else 'r04'
path = os.path.join(ROOT, 'DESIGN.md')
s = open(path).read()
for name, what in (('STATE_TABLE', 'eval'), ('BENCH_TABLE', 'bench'), ('TRAIN_TABLE', 'train')):
    out = subprocess.run([sys.executable, os.path.join(HERE, 'state_table.py'), rnd, what], capture_output=True, text=True)
    if out.returncode != 0:
        print(name, 'skipped:', out.stderr.strip().splitlines()[-1] if out.stderr.strip() else out.returncode)
        continue
    block = '<!-- %s -->\n%s<!-- /%s -->' % (name, out.stdout, name)
    pat = re.compile(r'<!-- %s -->.*?<!-- /%s -->' % (name, name), re.S)
    if pat.search(s):
        s = pat.sub(lambda m: block, s)
    else:
        s = s.replace('<!-- %s -->' % name, block)
open(path, 'w').write(s)
print('DESIGN.md tables refreshed from profiles/%s_*' % rnd)
