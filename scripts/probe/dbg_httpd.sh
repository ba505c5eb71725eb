cd $GRAFT_REPO_ROOT
python - <<'PY'
import struct, numpy as np, os
os.makedirs("/tmp/kd", exist_ok=True)
rows = np.array([[0.0],[1.0],[3.0]], dtype="<f4")
open("/tmp/kd/data.fbin","wb").write(struct.pack("<II",3,1)+rows.tobytes())
PY
vector_store_amd/vs_httpd --data-dir /tmp/kd --keyspace ks --index idx --metric l2sq --port 6123 --threads 2 > /tmp/httpd.out 2> /tmp/httpd.err &
PID=$!
sleep 4
python - <<'PY'
import http.client, json
try:
    c = http.client.HTTPConnection("127.0.0.1", 6123, timeout=5)
    c.request("GET", "/api/v1/indexes/ks/idx/status")
    r = c.getresponse(); print(r.status, r.read())
    c.request("POST", "/api/v1/indexes/ks/idx/ann", body=json.dumps({"vector":[0.0],"limit":3}), headers={"content-type":"application/json"})
    r = c.getresponse(); print(r.status, r.read())
except Exception as e:
    print("client error", repr(e))
PY
kill $PID; sleep 0.5; echo "--- stderr"; cat /tmp/httpd.err | grep -v amdgpu.ids | tail; ss -ltnp 2>/dev/null | head -5
