import CSiftmi
import Foundation

/// The reference's matchers are `static` functions of the value type (SIFT/SIFTDescriptor.swift:104, :298, :362) and its call sites
/// (Tests/SIFTMetalTests/DescriptorTests.swift:141-169: `SIFTDescriptor.match(source:target:absoluteThreshold:relativeThreshold:)`)
/// name no device.  They keep compiling unchanged: the statics below forward to ONE process-wide matcher context on
/// `SIFTMI355X.defaultDevice`, created at the first call.  A maintainer deletes the CPU bodies (SIFTDescriptor.swift:91-417: distance,
/// matchGeometry / compareGeometry, match, approximateMatch) from the value-type file; its stored properties and `init` stay.
public enum SIFTMI355X {
    /// HIP device ordinal of the process-wide matcher context; set it before the first matcher call.
    public static var defaultDevice: Int32 = 0

    static let matcher: OpaquePointer? = {
        var cfg = siftmi_config()
        siftmi_default_config(&cfg, 64, 64)                     // the matchers use none of the image-sized state
        cfg.n_octaves = 1
        var ctx: OpaquePointer? = nil
        let rc = siftmi_create(&cfg, defaultDevice, &ctx)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        return ctx
    }()
    /// a context is not re-entrant (include/siftmi.h): concurrent callers of the statics take turns
    static let lock = NSLock()

    static func pack(_ ds: [SIFTDescriptor]) -> [siftmi_descriptor] {
        ds.map { d in
            var r = siftmi_descriptor()
            r.theta = d.theta
            withUnsafeMutableBytes(of: &r.features) { p in
                for i in 0 ..< 128 { p[i] = UInt8(d.features[i]) }
            }
            return r
        }
    }

    static func coordinates(_ ds: [SIFTDescriptor]) -> [Float] {     // makeCoordinate, SIFTDescriptor.swift:146-151
        ds.flatMap { [$0.keypoint.absoluteCoordinate.x, $0.keypoint.absoluteCoordinate.y] }
    }

    typealias MatchCall = (OpaquePointer?, UnsafePointer<siftmi_descriptor>?, Int64, UnsafePointer<siftmi_descriptor>?, Int64, Int32,
                           Float, Float, UnsafeMutablePointer<UnsafePointer<siftmi_match>?>?, UnsafeMutablePointer<Int64>?) -> Int32

    static func correspondences(_ call: MatchCall, _ source: [SIFTDescriptor], _ target: [SIFTDescriptor],
                                _ absoluteThreshold: Float, _ relativeThreshold: Float) -> [SIFTCorrespondence] {
        let a = pack(source), b = pack(target)
        lock.lock(); defer { lock.unlock() }
        var out: UnsafePointer<siftmi_match>? = nil
        var n: Int64 = 0
        let rc = call(matcher, a, Int64(a.count), b, Int64(b.count), 0, absoluteThreshold, relativeThreshold, &out, &n)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        return (0 ..< Int(n)).map { i in                          // copied out before the lock goes: `out` is the context's buffer
            SIFTCorrespondence(source: source[Int(out![i].source)], target: target[Int(out![i].target)],
                               featureDistance: out![i].distance)
        }
    }
}

extension SIFTDescriptor {
    /// SIFTDescriptor.swift:298-318 -- brute force + ratio test on the int8 matrix cores (siftmi_match_descriptors)
    public static func match(source: [SIFTDescriptor], target: [SIFTDescriptor],
                             absoluteThreshold: Float = 1.176, relativeThreshold: Float = 0.6) -> [SIFTCorrespondence] {
        SIFTMI355X.correspondences(siftmi_match_descriptors, source, target, absoluteThreshold, relativeThreshold)
    }

    /// SIFTDescriptor.swift:362-388 -- the reference's ANN trie as sorted path codes (siftmi_approximate_match), same results
    public static func approximateMatch(source: [SIFTDescriptor], target: [SIFTDescriptor],
                                        absoluteThreshold: Float = 300, relativeThreshold: Float = 0.6) -> [SIFTCorrespondence] {
        SIFTMI355X.correspondences(siftmi_approximate_match, source, target, absoluteThreshold, relativeThreshold)
    }

    /// SIFTDescriptor.swift:104-144 -- match, then the geometric-consistency score of the first 80 matches (siftmi_match_geometry)
    public static func matchGeometry(source: [SIFTDescriptor], target: [SIFTDescriptor],
                                     absoluteThreshold: Float = 1.176, relativeThreshold: Float = 0.6) -> Float {
        let a = SIFTMI355X.pack(source), b = SIFTMI355X.pack(target)
        let axy = SIFTMI355X.coordinates(source), bxy = SIFTMI355X.coordinates(target)
        SIFTMI355X.lock.lock(); defer { SIFTMI355X.lock.unlock() }
        var score: Float = 0
        var n: Int64 = 0
        let rc = siftmi_match_geometry(SIFTMI355X.matcher, a, axy, Int64(a.count), b, bxy, Int64(b.count),
                                       absoluteThreshold, relativeThreshold, &score, &n)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        return score
    }
}

extension SIFT {
    /// the same match on this object's own context (its device, no lock shared with other callers)
    public func match(source: [SIFTDescriptor], target: [SIFTDescriptor],
                      absoluteThreshold: Float = 1.176, relativeThreshold: Float = 0.6) -> [SIFTCorrespondence] {
        let a = SIFTMI355X.pack(source), b = SIFTMI355X.pack(target)
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
