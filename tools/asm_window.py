"""Memory / wait / branch instructions of a kernel around the first line matching a pattern.
    KEEP=/tmp/dev.s tools/kres.sh none; python tools/asm_window.py /tmp/dev.s k_agent_stepILi2ELb0 'global_load_lds' [before] [after] [all]"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
kern, pat = sys.argv[2], sys.argv[3]
before = int(sys.argv[4]) if len(sys.argv) > 4 else 100
after = int(sys.argv[5]) if len(sys.argv) > 5 else 900
show_all = len(sys.argv) > 6
start = [i for i, l in enumerate(lines) if kern in l and l.endswith('>:')][0]
end = start + 1
while end < len(lines) and not (lines[end].endswith('>:') and lines[end][:1] == '0'):
    end += 1
body = lines[start:end]
idx = [i for i, l in enumerate(body) if re.search(pat, l)]
print("kernel lines", len(body), "matches at", idx[:20])
i0 = idx[0]
for k, l in enumerate(body[max(0, i0 - before):i0 + after]):
    t = l.split('//')[0].strip()
    if show_all or re.search(r'global_|buffer_|scratch_|s_waitcnt|s_barrier|s_cbranch|s_branch|lds', t):
        print(max(0, i0 - before) + k, t)
