// CPU baseline stand-in for the reference's serial Java path, in the reference's own arithmetic:
// java.math.BigInteger field elements reduced with mod() after every operation (algebra/fields/Fp.java:38-92),
// Jacobian add-2007-bl / dbl-2009-l (algebra/curves/barreto_naehrig/BNG1.java:38-161), and the window loop of
// VariableBaseMSM.pippengerMSM (algebra/msm/VariableBaseMSM.java:134-188): c = L - L/3 with L = log2(n),
// ceil(254 / c) windows from the most significant one down, digit 0 skipped, running sums, c doublings
// between windows.  The reference itself cannot run on the benchmark box (no Spark jars, no build), so
// bench.py compiles and times THIS file when a JDK is present and labels it a restatement; its result is
// compared with the GPU's bytes.
//
//   javac SerialPippenger.java && java SerialPippenger <input-file> <n>
// input-file: n x 96 B bases (X|Y|Z, 32-byte little-endian, the JNI wire format) then n x 32 B scalars.
// prints: "<seconds> <affine x hex> <affine y hex> <z: 1 or 0>"
import java.io.DataInputStream;
import java.io.FileInputStream;
import java.math.BigInteger;

public class SerialPippenger {
  static final BigInteger Q = new BigInteger("21888242871839275222246405745257275088696311157297823662689037894645226208583");
  static final BigInteger ZERO = BigInteger.ZERO, ONE = BigInteger.ONE;

  static BigInteger add(BigInteger a, BigInteger b) { return a.add(b).mod(Q); }
  static BigInteger sub(BigInteger a, BigInteger b) { return a.subtract(b).mod(Q); }
  static BigInteger mul(BigInteger a, BigInteger b) { return a.multiply(b).mod(Q); }
  static BigInteger sqr(BigInteger a) { return a.multiply(a).mod(Q); }

  static final class P {
    final BigInteger x, y, z;
    P(BigInteger x, BigInteger y, BigInteger z) { this.x = x; this.y = y; this.z = z; }
    boolean isZero() { return z.signum() == 0; }
  }
  static final P INF = new P(ZERO, ONE, ZERO);

  static P twice(P p) {  // dbl-2009-l, a = 0
    if (p.isZero()) return p;
    BigInteger a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    BigInteger d = sub(sub(sqr(add(p.x, b)), a), c);
    d = add(d, d);
    BigInteger e = add(add(a, a), a), f = sqr(e);
    BigInteger x3 = sub(f, add(d, d));
    BigInteger c8 = add(c, c); c8 = add(c8, c8); c8 = add(c8, c8);
    BigInteger y3 = sub(mul(e, sub(d, x3)), c8);
    BigInteger yz = mul(p.y, p.z);
    return new P(x3, y3, add(yz, yz));
  }

  static P add(P p, P q) {  // add-2007-bl
    if (p.isZero()) return q;
    if (q.isZero()) return p;
    BigInteger z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    BigInteger u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    BigInteger s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    if (u1.equals(u2) && s1.equals(s2)) return twice(p);
    BigInteger h = sub(u2, u1), s2ms1 = sub(s2, s1);
    BigInteger i = sqr(add(h, h)), j = mul(h, i), r = add(s2ms1, s2ms1), v = mul(u1, i);
    BigInteger x3 = sub(sub(sqr(r), j), add(v, v));
    BigInteger s1j = mul(s1, j);
    BigInteger y3 = sub(mul(r, sub(v, x3)), add(s1j, s1j));
    BigInteger z3 = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return new P(x3, y3, z3);
  }

  static BigInteger le(byte[] buf, int off) {
    byte[] be = new byte[33];
    for (int k = 0; k < 32; k++) be[32 - k] = buf[off + k];
    return new BigInteger(be);
  }

  public static void main(String[] args) throws Exception {
    final int n = Integer.parseInt(args[1]);
    byte[] bb = new byte[n * 96], sb = new byte[n * 32];
    try (DataInputStream in = new DataInputStream(new FileInputStream(args[0]))) {
      in.readFully(bb);
      in.readFully(sb);
    }
    P[] bases = new P[n];
    BigInteger[] scalars = new BigInteger[n];
    for (int i = 0; i < n; i++) {
      bases[i] = new P(le(bb, 96 * i), le(bb, 96 * i + 32), le(bb, 96 * i + 64));
      scalars[i] = le(sb, 32 * i);
    }
    final long t0 = System.nanoTime();
    final int log2 = Math.max(1, (int) (Math.log(n) / Math.log(2)));   // common/MathUtils.java:8-10
    final int c = log2 - (log2 / 3);
    final int numBuckets = 1 << c, numBits = 254, numGroups = (numBits + c - 1) / c;
    P result = INF;
    for (int k = numGroups - 1; k >= 0; k--) {
      P[] buckets = new P[numBuckets];
      for (int i = 0; i < numBuckets; i++) buckets[i] = INF;
      for (int i = 0; i < n; i++) {
        int id = 0;
        for (int j = 0; j < c; j++) if (scalars[i].testBit(k * c + j)) id |= 1 << j;
        if (id == 0) continue;
        buckets[id] = add(buckets[id], bases[i]);
      }
      P running = INF;
      for (int i = numBuckets - 1; i > 0; i--) {
        running = add(running, buckets[i]);
        result = add(result, running);
      }
      if (k > 0) for (int i = 0; i < c; i++) result = twice(result);
    }
    final double secs = (System.nanoTime() - t0) * 1e-9;
    if (result.isZero()) {
      System.out.println(secs + " 0 1 0");
    } else {
      BigInteger zi = result.z.modInverse(Q), zi2 = sqr(zi);
      System.out.println(secs + " " + mul(result.x, zi2).toString(16) + " " + mul(result.y, mul(zi2, zi)).toString(16) + " 1");
    }
  }
}
