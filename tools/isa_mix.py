"""Instruction mix of one kernel in the assembly `make -C simple_raytracer_amd/csrc asm` leaves in /tmp."""
import sys, re, collections
src = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "trace_kernelILb1ELb0ELb1E"
m = re.search(r'^(_ZN3srt12' + pat + r'.*?):\n(.*?)\.Lfunc_end', src, re.S | re.M)
body = m.group(2)
ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
print(m.group(1), len(ins), "instructions")
c = collections.Counter()
for i in ins:
    if 'f64' in i: c['f64'] += 1
    elif i.startswith('scratch_'): c[i[:13]] += 1
    elif i.startswith('ds_'): c['ds'] += 1
    elif i.startswith(('s_load', 's_buffer')): c['s_load'] += 1
    elif i.startswith('s_waitcnt'): c['waitcnt'] += 1
    elif i.startswith('global_'): c['global'] += 1
    elif i.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')): c['imul'] += 1
    elif i.startswith(('v_rcp', 'v_sqrt', 'v_rsq', 'v_div', 'v_exp', 'v_log')): c['trans/div'] += 1
    elif i.startswith(('v_readlane', 'v_writelane')): c['sgpr_spill_lane'] += 1
    elif i.startswith('v_'): c['valu'] += 1
    elif i.startswith(('s_cbranch', 's_branch')): c['branch'] += 1
    elif i.startswith('s_'): c['salu'] += 1
    else: c[i] += 1
print(c.most_common())
