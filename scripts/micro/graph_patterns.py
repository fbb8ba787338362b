"""Which multi-stream capture patterns does this ROCm / torch build survive?  Each pattern runs in its own process
(a bad one segfaults in hipStreamEndCapture).  Usage: python scripts/micro/graph_patterns.py [pattern]"""
import subprocess
import sys

PATTERNS = ['single', 'fork_join', 'fork_join_twice', 'two_sides', 'nested', 'event_join', 'event_pull', 'late_join_all',
            'record_stream', 'rejoin_joined', 'autograd_side', 'memset_side', 'wait_twice', 'nested_join_both', 'nested_prefork', 'nested_join_main_only',
            'nested_prefork_3', 'side_waits_side', 'side_waits_side_nowork', 'via_origin', 'side_waits_side_empty_join', 'mutual_via_origin', 'mutual_via_helpers', 'cycle3', 'nested_via_origin']


def run(p):
    import torch
    dev = torch.device('cuda:0')
    a = torch.ones(1 << 20, device=dev)
    s1, s2, s3 = (torch.cuda.Stream() for _ in range(3))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m = torch.cuda.current_stream()
        b = a * 2
        if p == 'fork_join':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            m.wait_stream(s1)
            d = c + b
        elif p == 'fork_join_twice':
            for _ in range(2):
                s1.wait_stream(m)
                with torch.cuda.stream(s1):
                    c = b + 1
                m.wait_stream(s1)
                b = c + b
        elif p == 'two_sides':
            s1.wait_stream(m); s2.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            with torch.cuda.stream(s2):
                d = b + 2
            m.wait_stream(s1); m.wait_stream(s2)
            e = c + d
        elif p == 'nested':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
                s2.wait_stream(s1)
                with torch.cuda.stream(s2):
                    d = c + 2
                s1.wait_stream(s2)
                e = d + c
            m.wait_stream(s1)
            f = e + b
        elif p in ('nested_join_both', 'nested_prefork', 'nested_join_main_only', 'nested_prefork_3'):
            if p.startswith('nested_prefork'):
                s2.wait_stream(m)                       # every sibling enters the capture from the ORIGIN stream first
                if p == 'nested_prefork_3':
                    s3.wait_stream(m)
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
                s2.wait_stream(s1)
                with torch.cuda.stream(s2):
                    d = c + 2
                    if p == 'nested_prefork_3':
                        s3.wait_stream(s2)
                        with torch.cuda.stream(s3):
                            d2 = d + 1
                        s2.wait_stream(s3)
                        d = d + d2
                if p != 'nested_join_main_only':
                    s1.wait_stream(s2)
                    e = d + c
                else:
                    e = c + 1
            m.wait_stream(s1)
            if p in ('nested_join_both', 'nested_join_main_only'):
                m.wait_stream(s2)
            if p == 'nested_prefork_3':
                m.wait_stream(s3); m.wait_stream(s2)
            f = e + b
        elif p in ('side_waits_side', 'side_waits_side_nowork', 'via_origin', 'side_waits_side_empty_join'):
            s1.wait_stream(m); s2.wait_stream(m)
            with torch.cuda.stream(s2):
                d = b + 2
            with torch.cuda.stream(s1):
                c = b + 1 if p != 'side_waits_side_nowork' else b
            if p == 'via_origin':
                m.wait_stream(s2)
                s1.wait_stream(m)
            else:
                s1.wait_stream(s2)
            with torch.cuda.stream(s1):
                e = c + d
            m.wait_stream(s1)
            if p != 'side_waits_side_empty_join':
                m.wait_stream(s2)
            f = e + b
        elif p in ('mutual_via_origin', 'mutual_via_helpers', 'cycle3', 'nested_via_origin'):
            def edge(src, dst):                          # dst waits for what src has queued
                if p == 'mutual_via_helpers':
                    h = torch.cuda.Stream()
                    h.wait_stream(src); dst.wait_stream(h)
                    keep.append(h)
                elif p == 'cycle3':
                    dst.wait_stream(src)
                else:
                    m.wait_stream(src); dst.wait_stream(m)
            keep = []
            if p == 'nested_via_origin':
                s1.wait_stream(m)
                with torch.cuda.stream(s1):
                    c = b + 1
                edge(s1, s2)                             # s1 forks s2 ...
                with torch.cuda.stream(s2):
                    d = c + 2
                with torch.cuda.stream(s1):
                    c2 = c * 2
                edge(s2, s1)                             # ... and joins it back
                with torch.cuda.stream(s1):
                    e = d + c2
                m.wait_stream(s1); m.wait_stream(s2)
            else:
                s1.wait_stream(m); s2.wait_stream(m); s3.wait_stream(m)
                with torch.cuda.stream(s1):
                    c = b + 1
                edge(s1, s2)
                with torch.cuda.stream(s2):
                    d = c + 2
                if p == 'cycle3':
                    edge(s2, s3)
                    with torch.cuda.stream(s3):
                        d = d + 1
                    edge(s3, s1)
                else:
                    edge(s2, s1)
                with torch.cuda.stream(s1):
                    e = d + c
                for h in keep:
                    m.wait_stream(h)
                m.wait_stream(s1); m.wait_stream(s2); m.wait_stream(s3)
            f = e + b
        elif p == 'event_join':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
                ev = torch.cuda.Event(); ev.record(s1)
            m.wait_event(ev)
            d = c + b
            m.wait_stream(s1)
        elif p == 'event_pull':                       # a stream enters the capture through an event wait, then is joined
            ev = torch.cuda.Event(); ev.record(m)
            s1.wait_event(ev)
            with torch.cuda.stream(s1):
                c = b + 1
            m.wait_stream(s1)
            d = c + b
        elif p == 'late_join_all':
            s1.wait_stream(m); s2.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            with torch.cuda.stream(s2):
                d = b + 2
            e = b * 3
            for s in (s1, s2, s3):
                with torch.cuda.stream(s):
                    cap = torch.cuda.is_current_stream_capturing()
                if cap:
                    m.wait_stream(s)
        elif p == 'record_stream':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            c.record_stream(m)
            m.wait_stream(s1)
            d = c + b
            del c
            e = d * 2
        elif p == 'rejoin_joined':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            m.wait_stream(s1)
            d = c + b
            m.wait_stream(s1)                          # a second join of the same, already joined stream
            e = d + 1
        elif p == 'wait_twice':
            s1.wait_stream(m)
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = b + 1
            m.wait_stream(s1)
        elif p == 'autograd_side':
            w = torch.ones(1 << 20, device=dev, requires_grad=True)
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                y = (w * b).sum()
            m.wait_stream(s1)
            z = (w * 3).sum() + y
            z.backward()
        elif p == 'memset_side':
            s1.wait_stream(m)
            with torch.cuda.stream(s1):
                c = torch.zeros(1 << 20, device=dev)
                c += b
            m.wait_stream(s1)
    g.replay()
    torch.cuda.synchronize()
    print('OK', p, flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, __file__, p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            tail = (r.stderr.strip().splitlines() or [''])[-1][:160]
            print('%-18s rc=%d %s %s' % (p, r.returncode, r.stdout.strip(), '' if r.returncode == 0 else tail), flush=True)
