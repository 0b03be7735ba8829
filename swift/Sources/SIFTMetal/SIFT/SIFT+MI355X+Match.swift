import CSiftmi

extension SIFT {
    public func match(source: [SIFTDescriptor], target: [SIFTDescriptor],
                      absoluteThreshold: Float = 1.176, relativeThreshold: Float = 0.6) -> [SIFTCorrespondence] {
        func pack(_ ds: [SIFTDescriptor]) -> [siftmi_descriptor] {
            ds.map { d in
                var r = siftmi_descriptor()
                r.theta = d.theta
                withUnsafeMutableBytes(of: &r.features) { p in
                    for i in 0 ..< 128 { p[i] = UInt8(d.features[i]) }
                }
                return r
            }
        }
        let a = pack(source), b = pack(target)
        var out: UnsafePointer<siftmi_match>? = nil
        var n: Int64 = 0
        let rc = siftmi_match_descriptors(ctx, a, Int64(a.count), b, Int64(b.count), 0,
                                          absoluteThreshold, relativeThreshold, &out, &n)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        return (0 ..< Int(n)).map { i in
            SIFTCorrespondence(source: source[Int(out![i].source)], target: target[Int(out![i].target)],
                               featureDistance: out![i].distance)
        }
    }
}
