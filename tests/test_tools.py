"""CPU tests of the analysis tools that read files (no GPU): tools/launch_gap.py --read on a synthetic kernel trace."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launch_gap_reads_a_kernel_trace(tmp_path):
    """Twelve march launches 500 us long and 17 us apart on queue 4, a sort of 60 us on queue 3 behind each, and one launch after a
    host synchronisation (3 ms later: not part of the steady state)."""
    d = tmp_path / "kt" / "host" / "1234_kernel_trace.csv"
    d.parent.mkdir(parents=True)
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1_000_000
    for k in range(12):
        rows.append(f'KERNEL_DISPATCH,1,4,"void vr::march_p2_kernel<1, true, false, false>(vr::MarchBatch, vr::PwQueue)",{t},{t + 500_000}')
        rows.append(f'KERNEL_DISPATCH,1,3,"vr::order_blocks_kernel(unsigned long long const*, int)",{t + 501_000},{t + 561_000}')
        t += 517_000
    t += 3_000_000
    rows.append(f'KERNEL_DISPATCH,1,4,"void vr::march_p2_kernel<1, true, false, false>(vr::MarchBatch, vr::PwQueue)",{t},{t + 500_000}')
    d.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_gap.py"), "--read", str(tmp_path / "kt")],
                         capture_output=True, text=True, check=True).stdout
    assert "11 back-to-back march launches" in out
    assert "kernel duration (tracer): median 500.0 us" in out
    assert "end -> next start: median 17.0 us" in out
    assert "vr::order_blocks_kernel: 11 launches, 60.0 us each, queues ['3']" in out
    assert "queues of the march launches: ['4']" in out
