import torch, ctypes
print('range', torch.cuda.Stream.priority_range())
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p)
        print('req', p, 'got', s.priority)
    except Exception as e:
        print('req', p, 'err', e)
hip = ctypes.CDLL('libamdhip64.so')
lo, hi = ctypes.c_int(0), ctypes.c_int(0)
print('hipDeviceGetStreamPriorityRange rc', hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), 'least', lo.value, 'greatest', hi.value)
for p in (lo.value, 0, hi.value):
    st = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), 0, p)
    q = ctypes.c_int(99)
    hip.hipStreamGetPriority(st, ctypes.byref(q))
    es = torch.cuda.ExternalStream(st.value)
    print('hip prio', p, 'rc', rc, 'got', q.value, 'external', es)
